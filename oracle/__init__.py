"""CPU oracle for the SparseFusion hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package; nothing under sparsefusion_amd/ does (tests/test_boundary.py greps for it).

  ngp_ref.c      plain-C restatement of the reference's CUDA-only entry points
                 (grid encode fwd/bwd, near_far_from_aabb, morton, packbits)
  ngp_native.py  ctypes/numpy binding of ngp_ref.c (+ torch autograd glue)
  ngp_ref.py     torch-CPU restatement of NeRFNetwork.common_forward / NeRFRenderer.run
  unet_ref.py    torch-CPU restatement of Unet.forward (functional, reference state-dict keys)
  plms_ref.py    restatement of the continuous-time schedule + PLMS sampler
  ref_loader.py  imports the REAL reference from /root/reference (dev container only) to
                 pin the restatements and generate tests/golden/*.pt
"""
