"""CPU oracle for the cine seg+flow hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it,
and only as the checker / the timed CPU baseline.  The product package
(cardiac-segmentation-optical-flow_amd/cineflow) never imports it and has no
CPU fallback: it fails loudly when libcineflow_hip.so is missing.

It restates, in plain PyTorch-CPU / numpy, the algorithms of the reference's
hot path, each function citing the reference file:line it
follows (paths relative to /root/reference).  Module classes in
oracle/models.py keep the reference's attribute names, so a reference
``state_dict`` loads with ``strict=True``.

Pinning status (see DESIGN.md "Oracle"):
  * pinned by the reference's own known-answer test:
      compute_steps_for_sliding_window  (tests/test_steps_for_sliding_window_prediction.py:96-163)
  * pinned by outputs of the reference classes run in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz):
      SpatialTransformer, VecInt, warp_linear, ConvGRUCell, DoubleConv/SingleConv blocks,
      Encoder2D, EncoderMotionAppearance, Decoder2D, TransformerFlowLayer / CrossAttentionLayer,
      TransformerFlowEncoderSuccessiveNoEmb, PositionEmbeddingSine2d, Generic_UNet,
      SegFlowGaussian (motion_appearance dispatch), OpticalFlowModelSuccessive + ModelWrap,
      _get_gaussian, TTA mirroring, jacobian_determinant (numpy restatement of
      compute_jacobian.py, which needs the absent pystrum only for a meshgrid).
  * PARITY UNPINNED (source absent from the reference snapshot; restated from the
    call sites + the published RAFT definition):
      CorrVolume, CorrBlock, BasicUpdateBlock, coords_grid, and therefore the
      video.yaml dispatch (forward_..._cost_volume_transformer_cat) and the RAFT loop;
      pad_nd_image (batchgenerators, un-vendored) and NormalizeIntensity (monai, absent).
"""
