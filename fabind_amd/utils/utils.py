"""The two helpers the reference's models/model.py imports from utils/utils.py (lines 150-158, 687-699)."""
import torch


def get_keepNode_tensor(protein_node_xyz, pocket_radius, add_noise_to_com, chosen_pocket_com):
    """Residues strictly closer than `pocket_radius` (Angstrom) to the pocket centre.

    `chosen_pocket_com` is one centre [3] (reference usage) or one centre per residue [n,3] (batched use)."""
    if add_noise_to_com:
        chosen_pocket_com = chosen_pocket_com + add_noise_to_com * (2 * torch.rand_like(chosen_pocket_com) - 1)
    if chosen_pocket_com.dim() == 1:
        chosen_pocket_com = chosen_pocket_com.unsqueeze(0)
    dis = torch.sqrt(torch.sum((protein_node_xyz - chosen_pocket_com) ** 2, dim=-1))
    return dis < pocket_radius


def gumbel_softmax_no_random(logits, tau=1, hard=False, eps=1e-10, dim=-1):
    """softmax(logits / tau) -- the noise-free Gumbel-softmax used in eval; straight-through if `hard`."""
    y_soft = (logits / tau).softmax(dim)
    if not hard:
        return y_soft
    index = y_soft.max(dim, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(dim, index, 1.0)
    return y_hard - y_soft.detach() + y_soft
