from yogo_amd.utils.prediction_formatting import (  # noqa: F401
    count_cells_for_formatted_preds,
    format_preds,
    PredictionLabelMatch,
    format_preds_and_labels_v2,
    format_preds_and_labels_v2_batched,
    format_preds_batched,
    format_to_numpy,
    format_to_numpy_batched,
    get_prediction_class_counts,
    prediction_rows_to_text,
    save_predictions,
    split_batched,
)
