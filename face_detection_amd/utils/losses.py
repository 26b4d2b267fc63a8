"""MI355X stand-in for the forward and backward pass of reference utils/losses.py (SSDLoss).

Same constructor arguments and the same per-image loss vector as SSDLoss.call
(utils/losses.py:83-181); the gradient with respect to y_pred, which the reference leaves to
TensorFlow's autodiff, is returned explicitly by call_with_grad."""
from ..batched import ssd_loss_batch


class SSDLoss:
    def __init__(self, neg_pos_ratio=3, n_neg_min=0, alpha=1.0):
        self.neg_pos_ratio = int(neg_pos_ratio)
        self.n_neg_min = int(n_neg_min)
        self.alpha = float(alpha)

    def call(self, y_true, y_pred):
        """(B,) loss per image, utils/losses.py:97-181."""
        return ssd_loss_batch(y_true, y_pred, self.neg_pos_ratio, self.n_neg_min, self.alpha)

    __call__ = call

    def call_with_grad(self, y_true, y_pred):
        """(loss (B,), d mean(loss) / d y_pred (B,A,6))."""
        return ssd_loss_batch(y_true, y_pred, self.neg_pos_ratio, self.n_neg_min, self.alpha,
                              return_grad=True)

    def get_config(self):
        return {"neg_pos_ratio": self.neg_pos_ratio, "n_neg_min": self.n_neg_min, "alpha": self.alpha}
