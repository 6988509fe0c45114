"""forgex_amd -- MI355X-native batch regex matching behind Forgex's `use forgex` API
(`.in.`, `.match.`, `regex`, `regex_f`, `is_valid_regex`).  See DESIGN.md / INTEGRATION.md."""
from ._lib import build, lib, LIB_PATH, OP_SEARCH, OP_MATCH  # noqa: F401
from .api import Batch, Program, in_, match, regex, regex_f, is_valid_regex, strerror, match_many, packed_layout, unpack_results, pinned, INVALID_CHAR_INDEX  # noqa: F401
