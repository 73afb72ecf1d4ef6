from .local_consistent_loss import LocalConsistentLoss  # noqa: F401
from .negative_learning_loss import NegativeLearningLoss  # noqa: F401
