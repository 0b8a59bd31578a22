"""MI355X-native drop-in for the reference's `fabind/models` package (same class names, constructor
signatures, parameter names/shapes -> identical `state_dict` keys, same forward signatures)."""
from .model import get_model, IaBNet_mean_and_pocket_prediction_cls_coords_dependent  # noqa: F401
from .att_model import EfficientMCAttModel, ComplexGraph  # noqa: F401
from .egnn import MCAttEGNN, MC_E_GCL, MC_Att_L, FABindLayer  # noqa: F401
from .cross_att import CrossAttentionModule, RowAttentionBlock  # noqa: F401
from .model_utils import Attention, Transition, InteractionModule  # noqa: F401
