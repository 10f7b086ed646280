"""dcz_amd -- MI355X-native hot path of the DataComp chunked canonical-Huffman compressor.

The directory name is the one the build contract fixes; it is not an importable identifier, so load it
with `__graft_entry__.load_package()` (registers it as module `dcz_amd`).
"""
from . import container, native  # noqa: F401
from .native import Context, DczError, lib  # noqa: F401
from .service import (DeviceBlocks, HipCompressionService, HipFrequencyService, HuffmanDecodeError,  # noqa: F401
                      StageMetrics)
