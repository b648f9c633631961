"""Host-side data helpers on either side of the hot path: evaluation of the detections the model emits."""
