"""piccolo_amd — MI355X (gfx950) implementation of PICCOLO's sampling-loss hot path.

    piccolo_amd.omniloc / utils / parse_utils / localize   the reference's Python surface (same names and semantics)
    piccolo_amd.ops                                          torch-tensor front end of the C ABI (include/piccolo_hip.h)
    piccolo_amd.dist                                         query images sharded over GPUs, one gather of the results
    piccolo_amd.synth                                        seeded synthetic rooms for tests and bench
    piccolo_amd.build                                        hipcc build of csrc/*.hip -> lib/libpiccolo_hip.so

Every computation runs in hand-written HIP kernels; there is no CPU fallback (ops raise without the library or a GPU).
"""
__version__ = "0.1.0"
