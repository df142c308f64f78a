from yogo_amd.utils.prediction_formatting import (  # noqa: F401
    count_cells_for_formatted_preds,
    format_preds,
    format_preds_batched,
    get_prediction_class_counts,
    split_batched,
)
