"""Randomised constructor arguments (the reference's kwargs: vit.py:107-108, deepvit.py:113-114, cait.py:150-151), random batch: forward + full
backward through the C ABI against the oracle on identical weights.  A fixed-seed slice of tools/fuzz_configs.py (rectangular images and patches,
one head, to_out as the identity, mean pooling, widths that are not multiples of 4, one- and two-token images ...)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compute,n,seed", [("fp32", 30, 0), ("bf16x3", 16, 2), ("bf16", 30, 1)])
def test_random_configurations_match_the_oracle(compute, n, seed):
    import fuzz_configs
    fails = fuzz_configs.run(n, seed, compute)
    assert not fails, fails


@pytest.mark.parametrize("compute,n,seed", [("fp32", 16, 11), ("bf16x3", 12, 11), ("bf16", 24, 11)])
def test_token_counts_around_the_attention_dispatch_boundaries(compute, n, seed):
    """63 .. 400 tokens per image: the 4 / 6 / 14 / 18-tile instantiations of the fused attention kernels, the materialised path past 288 tokens, the
    64-key sweeps of the DeepViT / CaiT head-axis kernels."""
    import fuzz_configs
    fails = fuzz_configs.run(n, seed, compute, "tokens")
    assert not fails, fails


@pytest.mark.parametrize("compute,n,seed", [("fp32", 20, 31), ("bf16", 20, 31)])
def test_random_sibling_models_match_the_oracle(compute, n, seed):
    """parallel_vit.ViT with 2-3 branches and vit_with_patch_merger.ViT with a random merge layer / token count"""
    import fuzz_configs
    fails = fuzz_configs.run(n, seed, compute, "siblings")
    assert not fails, fails


@pytest.mark.parametrize("compute,n,seed", [("fp32", 12, 41), ("bf16", 20, 41)])
def test_call_sequences_on_one_handle_match_the_oracle(compute, n, seed):
    """One handle per random configuration, five calls with a changing batch / a smaller image / new weights / two backward passes on one forward,
    every call against the oracle.  (Round 6: this sweep found that a bf16 CaiT handle computed its patch-stage attention from zeroed [to_q | to_kv]
    operand copies after the first change of geometry -- profiles/r6/fuzz_sequences_bf16_BEFORE_the_qkvcat_fix_r6as.log.)"""
    import fuzz_configs
    fails = fuzz_configs.run_sequences(n, seed, compute)
    assert not fails, fails


@pytest.mark.parametrize("compute,n,seed", [("fp32", 6, 61), ("bf16", 10, 61)])
def test_wide_models_match_the_oracle(compute, n, seed):
    """dim 768 .. 4096, 8 .. 32 heads, mlp_dim up to 8192 on a handful of tokens: every LayerNorm row form, the head-axis kernels at 24 / 32 heads"""
    import fuzz_configs
    fails = fuzz_configs.run(n, seed, compute, "wide")
    assert not fails, fails


def test_call_sequences_at_the_shapes_the_big_kernels_take():
    """256 .. 768 wide, 101 .. 257 tokens, up to 24 images, changing batch / image size / weights on one bf16 handle: thousands of token rows, i.e. the
    pipelined persistent GEMM with its fused epilogues, the fused attention kernels at 14 / 18 key tiles, split-K weight gradients."""
    import fuzz_configs
    fails = fuzz_configs.run_sequences(8, 101, "bf16", steps=3, medium=True)
    assert not fails, fails


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_random_mae_and_simmim_wrappers_match_the_oracle(compute):
    """random encoders, image / patch sizes, masking ratios from one masked patch to all but one, decoder widths (tools/fuzz_wrappers.py)"""
    import fuzz_wrappers
    fails = fuzz_wrappers.run(16, 0, compute)
    assert not fails, fails


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_random_t2t_vits_match_the_oracle(compute):
    """random tokenizer layers / image sizes (tf.image.extract_patches 'SAME' geometries, tokenizer transformers of odd widths), two batch sizes per object
    (tools/fuzz_t2t.py; configurations the reference's own size formula rejects are skipped)"""
    import fuzz_t2t
    fails = fuzz_t2t.run(40, 2, compute)
    assert not fails, fails
