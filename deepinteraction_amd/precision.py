"""Precision policy of fp16 inference on this path (round 3).

`half_maps_(encoder, decoder)` is how a model is put into the benched fp16 mode: everything that works on feature MAPS -
the whole MMRI encoder, the decoder's two heat-map heads - goes to fp16 (fp32 accumulation in every kernel); everything
on the B*Q query TOKENS of the MMPI decoder - decoder layer, RoI blocks, DynamicConv, prediction heads, positional
embeddings, class encoding, the cross attention's K / V projection weight - keeps its float32 parameters and runs on the
float32 token kernels (csrc/token32.hip, csrc/cross_attn.hip).

Why (tests/tools/fp16_error_budget.py, oracle at shape R, proposals forced equal): rounding ONLY the decoder layer's
weights to fp16 moves 24-32 % of the box outputs by more than 1e-3 of their scale (the 200 x 32 400 cross attention has
logits of magnitude ~500); the K/V projection weight alone 5-8 %; the prediction heads alone 4 %; K or q of the cross
attention in fp16 5 %.  The token path is < 1 % of the forward's bytes and flops, so float32 there costs nothing, while
`module.half()` on the whole head (still accepted: the fused path widens fp16 parameters) gives exactly those errors.
"""
import torch


def half_maps_(encoder, decoder):
    """In place: fp16 for the map side, float32 for the token side.  Returns (encoder, decoder)."""
    encoder.half()
    decoder.float()
    decoder.heatmap_head.half()
    decoder.heatmap_head_img.half()
    return encoder, decoder


def to_inference(encoder, decoder, dtype):
    """fp16: `half_maps_`; float32: everything float32."""
    if dtype == torch.float16:
        return half_maps_(encoder, decoder)
    return encoder.to(dtype), decoder.to(dtype)
