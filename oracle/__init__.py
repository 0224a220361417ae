"""CPU oracle for the RadarDistill training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain PyTorch-CPU / numpy / C restatement of the reference algorithm (yyongjae/RadarDistill,
an OpenPCDet fork) for every row of SURVEY.md section 8(a).  Each function cites the reference
file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package, and only as the checker / the timed CPU baseline.  The product
(`radardistill_amd`) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * pinned by goldens generated from reference leaf modules imported in the build container
    (tests/golden/make_golden.py): dynamic pillar VFE (modulo torch_scatter arithmetic),
    BaseBEVBackboneV2, ConvNeXtBlock/LayerNorm/GRN, Radar_Distill.forward/low_loss/high_loss/
    get_loss, Radar_CenterHead.forward/assign_targets, focal / L1 / DIoU losses, gaussian radius.
  * pinned by the reference's own known-answer identities (pcdet/ops/basicblock/test.py):
    DCNv2 zero-offset+unit-mask == nn.Conv2d, gradcheck.
  * PARITY UNPINNED (third-party binaries absent, no reference tests): spconv SubMConv2d /
    SparseConv2d / dense(), torch_scatter mean/max, iou3d rotated overlap.  Restated from
    documented semantics / the in-tree CUDA source; checked with analytic known answers.
"""
