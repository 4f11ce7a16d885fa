"""`from tensorflow.python import pywrap_tensorflow` (utils/general.py:21 of the reference) -- shim, tests only."""


class _PywrapTensorflow(object):
    @staticmethod
    def NewCheckpointReader(path):
        raise NotImplementedError("TF checkpoints are not readable through the NumPy shim")


pywrap_tensorflow = _PywrapTensorflow()
