"""MI355X-native EM refinement hot path of BaMMmotif2.

`csrc/` holds the HIP kernels and the C ABI (include/bamm_em.h); `em.py` mirrors the
reference's EM / ScoreSeqSet interface over that ABI; `synth.py` makes the benchmark inputs.
"""
from . import abi, synth  # noqa: F401
from .em import (EM, Comm, Context, PackedSeqs, SeqSet, calculate_p, logodds, seed_from_pwm, sample_negatives, libc_srand,  # noqa: F401
                 v_size, v_offset, bg_size, bg_offset, device_count, device_pci_bus_id, device_can_access_peer, peer_access_matrix)

__all__ = ["EM", "Comm", "Context", "PackedSeqs", "SeqSet", "calculate_p", "logodds", "seed_from_pwm", "sample_negatives", "libc_srand",
           "v_size", "v_offset", "bg_size", "bg_offset", "device_count", "device_pci_bus_id", "device_can_access_peer", "peer_access_matrix", "abi", "synth"]
