"""CPU: bert_vits2_amd/schema.py names every tensor the reference's inference sub-networks hold, with the same shapes
(fixture written by oracle/gen_golden.py from the reference's own state_dict())."""
import json
import os

from bert_vits2_amd import hparams as H, schema
from tests.helpers import GOLDEN


def test_schema_matches_reference_state_dict():
    ref = json.load(open(os.path.join(GOLDEN, "reference_state_dict_schema.json")))
    for tag, tf in (("transformer_flow", True), ("residual_flow", False)):
        ours = {k: list(v) for k, v in schema.param_shapes(H.default_v23(use_transformer_flow=tf)).items()}
        assert ours == ref[tag], (set(ours) ^ set(ref[tag]))


def test_param_count():
    assert schema.n_params(H.default_v23()) == 49885433
