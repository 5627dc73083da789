"""Look-alikes of the two detectron2 containers the reference's callers touch
(``Instances`` / ``Boxes``; eval/refiner_model.py:267-271 uses ``output['instances'].to('cpu').pred_masks``)."""
import torch


class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def __len__(self):
        return self.tensor.shape[0]


class Instances:
    def __init__(self, image_size, **fields):
        self._image_size = tuple(image_size)
        self._fields = {}
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def __getattr__(self, name):
        if name.startswith("_") or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def to(self, *args, **kwargs):
        out = Instances(self._image_size)
        for k, v in self._fields.items():
            out.set(k, v.to(*args, **kwargs) if hasattr(v, "to") else v)
        return out
