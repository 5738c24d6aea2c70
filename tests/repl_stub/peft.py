"""Stand-in for the `peft` package, which this image does not have: the reference's REPL imports `PeftModel` at module level
(scripts/inference/inference.py:3) and only uses it on the un-merged --lora_model path (:66-75).  TEST INFRASTRUCTURE (tests/test_reference_repl.py)."""


class PeftModel:
    @classmethod
    def from_pretrained(cls, model, model_id, *args, **kwargs):
        return model          # the package folds the adapter at load time (weights.fold_lora); nothing left to wrap


class PeftMixedModel(PeftModel):      # transformers probes for it when a `peft` module is importable
    pass


__version__ = "0.0.0"
