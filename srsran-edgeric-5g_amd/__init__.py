"""MI355X-native 5G NR downlink PHY hot path (PDSCH processor + OFDM modulator).

Host-side Python plumbing over the C ABI of include/mi355_nrphy.h.  The compute lives in the HIP
library ``csrc/libmi355nrphy.so`` (built by ``build.py`` / ``__graft_entry__.build()``); importing this
package never falls back to a CPU implementation -- ``lib.load()`` raises when the library is missing.
"""
from . import abi  # noqa: F401
from . import lib  # noqa: F401
from . import sharding  # noqa: F401

__all__ = ["abi", "lib", "sharding"]
