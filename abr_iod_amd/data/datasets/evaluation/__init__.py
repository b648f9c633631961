"""evaluate(dataset, predictions, output_folder, **kwargs) (mirror of maskrcnn_benchmark/data/datasets/evaluation/__init__.py):
only the PASCAL VOC protocol is on this path (every configs/voc YAML evaluates with it)."""
from .voc import voc_evaluation


def evaluate(dataset, predictions, output_folder, **kwargs):
    args = dict(dataset=dataset, predictions=predictions, output_folder=output_folder, **kwargs)
    return voc_evaluation(**args)
