"""CPU oracle for the DeepInteraction interaction hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain PyTorch-fp32 / numpy / C restatement of the reference algorithm
(`/root/reference/projects/mmdet3d_plugin/...`, cited function by function) and
of the third-party semantics the reference calls into (mmdet3d 0.17.1,
detectron2 ROIAlignV2, OpenCV morphology, torch grid_sample / MHA).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `deepinteraction_amd/` or
`projects/` imports it; the product path raises if the HIP library is absent.

Pinning status (see DESIGN.md "Oracle"):
  * the reference's own Python (encoder_utils.py, deepinteraction_encoder.py,
    decoder_utils.py, deepinteraction_decoder.py, transfusion_bbox_coder.py,
    depth_map_utils.py) is imported *as is* in the build container by
    `oracle/refpin/` with stubs for the absent third-party packages, and the
    restatement here is checked against it (`tests/test_oracle_vs_reference.py`,
    skipped when /root/reference is absent) and against golden vectors it
    generated (`tests/golden/`, produced by `oracle/refpin/make_golden.py`);
  * the reference's CUDA kernels (`locatt_ops/kernels.cuh`) are compiled for the
    host CPU from where they lie into `oracle/_ref/liblocatt_ref.so` by
    `oracle/Makefile` and the C restatement `oracle/locatt_c.c` is checked
    bit-for-bit against them;
  * the third-party boundaries (mmdet3d helpers, detectron2 ROIAlign, OpenCV)
    are restated from their published algorithms: PARITY UNPINNED there - the
    reference ships no tests or golden vectors.
"""
