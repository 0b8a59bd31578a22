from . import att_model, cross_att, egnn, model, model_utils  # noqa: F401
from .model import FABindPlus, get_model  # noqa: F401
