"""Optimizer configuration objects (the arithmetic runs in the HIP Adam kernel).

``Adam`` mirrors ``tf.keras.optimizers.Adam`` as the reference's notebooks construct it
(notebooks/Training.ipynb cells 4 and 10): positional learning rate, Keras defaults
beta_1 = 0.9, beta_2 = 0.999, epsilon = 1e-7.  The learning rate is a float32 variable
[K]: reading ``opt.lr`` returns the float32-rounded value, which is what makes the
ReduceLROnPlateau sequences of the notebooks reproducible.
"""
import numpy as np


class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False,
                 name="Adam", **kwargs):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference and is not implemented")
        if "lr" in kwargs:
            learning_rate = kwargs.pop("lr")
        self._lr = np.float32(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.name = name
        self.iterations = 0

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = np.float32(value)

    learning_rate = lr

    def get_config(self):
        return {"name": self.name, "learning_rate": float(self._lr), "decay": 0.0, "beta_1": self.beta_1,
                "beta_2": self.beta_2, "epsilon": self.epsilon, "amsgrad": False}


def get(identifier):
    if isinstance(identifier, Adam):
        return identifier
    if isinstance(identifier, str) and identifier.lower() == "adam":
        return Adam()
    raise ValueError("unsupported optimizer %r: the engine implements Adam (the reference's optimizer)" % (identifier,))
