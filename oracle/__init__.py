"""CPU oracle for the SparseFusion hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package; nothing under sparsefusion_amd/ does (tests/test_boundary.py greps for it).

  ngp_ref.c      plain-C restatement of the reference's CUDA-only entry points
                 (grid encode fwd/bwd, near_far_from_aabb, morton, packbits)
  ngp_native.py  ctypes/numpy binding of ngp_ref.c (+ torch autograd glue)
  build_ref.py   compiles the REFERENCE's gridencoder.cu / raymarching.cu for the host CPU through cuda_shim/ into
                 _ref/libref_native*.so (dev container; the built library travels to the GPU box)
  ref_native.py  ctypes binding of that library; ref_exports.cpp = its C-ABI doors; pins ngp_ref.c bit for bit
                 (tests/test_oracle_native_pin.py, tests/golden/ngp_native.pt)
  ngp_ref.py     torch-CPU restatement of NeRFNetwork.common_forward / NeRFRenderer.run
  unet_ref.py    torch-CPU restatement of Unet.forward (functional, reference state-dict keys)
  plms_ref.py    restatement of the continuous-time schedule + PLMS sampler
  ref_loader.py  imports the REAL reference from /root/reference (dev container only) to
                 pin the restatements and generate tests/golden/*.pt
"""
