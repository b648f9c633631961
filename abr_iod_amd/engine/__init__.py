from .smoke import smoke_step  # noqa: F401
from .trainer import do_train, reduce_loss_dict, train_step  # noqa: F401
