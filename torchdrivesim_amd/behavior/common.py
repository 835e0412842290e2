class InitializationFailedError(RuntimeError):
    """A scene could not be initialised from the given data (behavior/common.py)."""
