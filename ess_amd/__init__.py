"""ess_amd -- MI355X-native (gfx950) implementation of the uzh-rpg/ess hot path.

Host code mirrors the reference's class surface (ess_amd.e2vid.*, ess_amd.models.*, ess_amd.training.*);
all arithmetic runs in hand-written HIP kernels behind the C ABI of include/ess_hip.h.
"""
__version__ = '0.1.0'
