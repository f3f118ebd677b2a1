"""mazu_amd -- MI355X-native SSHash k-mer -> unitig dictionary behind mazu's K2U surface.

Only the hot path: C++ host library owning the index layout + hand-written gfx950 kernels
(mazu_amd/csrc), reached through the C ABI in include/mazu_amd.h.  This package is the thin
host-side mirror of the reference's `kphf` / `index` interface used by tests and bench.py.
"""
from ._capi import MZ_NO_SKEW_INDEX, MazuError, lib  # noqa: F401
from .kphf import CanonicalKmer, K2UPos, MatchType, SSHash, StreamingK2U, UnitigSet, K2UPOS_DTYPE, kmer_word  # noqa: F401
