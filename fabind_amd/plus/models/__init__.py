from . import att_model, cross_att, egnn, model, model_utils  # noqa: F401
from .model import FABindPlus, compute_loss, get_model  # noqa: F401
