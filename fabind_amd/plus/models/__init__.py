from . import att_model, cross_att, egnn, model_utils  # noqa: F401
