"""BoxList: the currency between the model's components (mirror of
maskrcnn_benchmark/structures/bounding_box.py:9-257 for the fields and methods the hot path touches).
Boxes are an [n,4] fp32 xyxy device tensor + image size (W,H) + a dict of per-box extra fields;
indexing a BoxList indexes every field (:205-209)."""
import torch


FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        dev = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=dev)
        if bbox.ndimension() != 2 or bbox.size(-1) != 4:
            raise ValueError("bbox should be [n,4], got {}".format(tuple(bbox.shape)))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox, self.size, self.mode = bbox, image_size, mode  # size = (image_width, image_height)
        self.extra_fields = {}

    # --- fields
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, other):
        self.extra_fields.update(other.extra_fields)

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        for f in ([fields] if not isinstance(fields, (list, tuple)) else fields):
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return out

    # --- geometry
    def convert(self, mode):
        if mode == self.mode:
            return self
        x1, y1, a, b = self.bbox.unbind(-1)
        if mode == "xywh":  # from xyxy, TO_REMOVE = 1 (:62-66)
            box = torch.stack((x1, y1, a - x1 + 1, b - y1 + 1), -1)
        else:               # from xywh (:75-83)
            box = torch.stack((x1, y1, x1 + (a - 1).clamp(min=0), y1 + (b - 1).clamp(min=0)), -1)
        out = BoxList(box, self.size, mode)
        out._copy_extra_fields(self)
        return out

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    def clip_to_image(self, remove_empty=True):
        w, h = self.size
        self.bbox[:, 0].clamp_(min=0, max=w - 1)
        self.bbox[:, 1].clamp_(min=0, max=h - 1)
        self.bbox[:, 2].clamp_(min=0, max=w - 1)
        self.bbox[:, 3].clamp_(min=0, max=h - 1)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def resize(self, size, *args, **kwargs):
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        box = self.convert("xyxy").bbox * torch.tensor([rw, rh, rw, rh], device=self.bbox.device)
        out = BoxList(box, size, "xyxy")
        for k, v in self.extra_fields.items():
            out.add_field(k, v if isinstance(v, torch.Tensor) else v.resize(size, *args, **kwargs))
        return out.convert(self.mode)

    def transpose(self, method):
        """FLIP_LEFT_RIGHT (0) / FLIP_TOP_BOTTOM (1) of the boxes (bounding_box.py:128-167; TO_REMOVE = 1 for the x flip only)"""
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        w, h = self.size
        b = self.convert("xyxy").bbox
        x1, y1, x2, y2 = b[:, 0:1], b[:, 1:2], b[:, 2:3], b[:, 3:4]
        if method == FLIP_LEFT_RIGHT:
            box = torch.cat((w - x2 - 1, y1, w - x1 - 1, y2), dim=-1)
        else:
            box = torch.cat((x1, h - y2, x2, h - y1), dim=-1)
        out = BoxList(box, self.size, "xyxy")
        for k, v in self.extra_fields.items():
            out.add_field(k, v if isinstance(v, torch.Tensor) else v.transpose(method))
        return out.convert(self.mode)

    # --- tensor-like
    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def __repr__(self):
        return "BoxList(num_boxes={}, image_width={}, image_height={}, mode={})".format(len(self), self.size[0], self.size[1], self.mode)
