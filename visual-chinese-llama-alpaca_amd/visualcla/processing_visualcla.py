"""VisualCLAProcessor: tokenizer + CLIP image processor behind one call.  The reference exports this wrapper
(models/visualcla/__init__.py:6, processing_visualcla.py) but no script uses it; kept as a thin host-side helper so
`from visualcla import VisualCLAProcessor` keeps working."""
from __future__ import annotations


class VisualCLAProcessor:
    attributes = ["image_processor", "tokenizer"]

    def __init__(self, image_processor=None, tokenizer=None, **kwargs):
        if image_processor is None:
            raise ValueError("You need to specify an `image_processor`.")
        if tokenizer is None:
            raise ValueError("You need to specify a `tokenizer`.")
        self.image_processor, self.tokenizer = image_processor, tokenizer

    def __call__(self, text=None, images=None, return_tensors=None, **kwargs):
        if text is None and images is None:
            raise ValueError("You have to specify either text or images. Both cannot be none.")
        enc = self.tokenizer(text, return_tensors=return_tensors, **kwargs) if text is not None else None
        feats = self.image_processor(images, return_tensors=return_tensors, **kwargs) if images is not None else None
        if enc is not None and feats is not None:
            enc["pixel_values"] = feats.pixel_values
            return enc
        return enc if enc is not None else feats

    def batch_decode(self, *args, **kwargs):
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.tokenizer.decode(*args, **kwargs)

    @property
    def model_input_names(self):
        names = list(getattr(self.tokenizer, "model_input_names", [])) + list(getattr(self.image_processor, "model_input_names", []))
        return list(dict.fromkeys(names))
