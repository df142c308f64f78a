"""Default hyper-parameters of the CLI (yogo/utils/default_hyperparams.py:1-12)."""


class DefaultHyperparams:
    BATCH_SIZE = 64
    EPOCHS = 64
    LEARNING_RATE = 3e-4
    LABEL_SMOOTHING = 0.01
    DECAY_FACTOR = 10
    WEIGHT_DECAY = 5e-2
    IOU_WEIGHT = 5.0
    NO_OBJ_WEIGHT = 0.5
    CLASSIFY_WEIGHT = 1.0
    ANCHOR_W = 0.04250100424705710   # (the reference's cluster result at full precision: these become model buffers)
    ANCHOR_H = 0.05551774140353888
