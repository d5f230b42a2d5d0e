"""CPU oracle for the CenterTrack per-frame inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the *checker* (never as the thing measured or
shipped).  The product path (``centertrack_amd``) never imports this package and
fails loudly when the HIP library is missing.

What is restated (every function cites the reference file:line it follows,
relative to the upstream checkout ``xingyizhou/CenterTrack``):

  dcn_v2.py / dcn_v2_ref.c  modulated deformable conv v2 forward.  The arithmetic
                            lives in the un-vendored, un-pinned submodule
                            ``CharlesShang/DCNv2`` (``.gitmodules:10-13``, branch
                            master, no SHA); its published algorithm is restated
                            (SURVEY.md Appendix B).  **PARITY UNPINNED for this op**:
                            the reference ships no test/golden vector for it and the
                            source is absent, so it is anchored on known-answer tests
                            (zero offset == conv2d, integer shift, mask == 0, border
                            cases) and on two independent restatements (vectorised
                            torch, scalar C) agreeing with each other.
  dla34.py                  DLA-34 + DLAUp + IDAUp + heads forward
                            (``src/lib/model/networks/dla.py``, ``base_model.py``).
  decode.py                 ``_nms``/``_topk``/``generic_decode``
                            (``src/lib/model/decode.py``, ``model/utils.py``).
  image.py                  affine transforms, Gaussian rendering (``utils/image.py``).
  post_process.py           ``generic_post_process`` (+ ddd helpers).
  tracker.py                ``Tracker`` + ``greedy_assignment`` (``utils/tracker.py``).
  detector.py               ``Detector.run``'s device-side part on CPU
                            (``src/lib/detector.py``).

Pinning: everything except DCNv2 is pinned against the reference's *own* Python
modules imported from ``/root/reference`` in the build container; the outputs
are committed under ``tests/golden/`` together with the generating script
``tests/golden/make_golden.py``.  The GPU box has no ``/root/reference``.
"""
