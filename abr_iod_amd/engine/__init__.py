from .trainer import do_train, reduce_loss_dict, train_step  # noqa: F401
