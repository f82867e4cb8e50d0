"""
Voice bank: batched rendering of structurally identical voice sub-graphs under a MixPE.

(Placeholder for the batched path; until it is enabled MixPE renders its inputs one by
one.  See DESIGN.md, "voice banks".)
"""

from __future__ import annotations


def try_build_bank(inputs):
    return None
