"""tools/ only: OI_LIB=ablation[_<tag>] makes a TOOL load the -DOI_ABLATION build (tools/build_ablation.sh) instead of
the product library.  The package loader itself ignores the variable (tests/ and bench.py always load
openintel_amd/libopenintel_hip.so); a tool opts in by importing this module right after `import openintel_amd`."""
import os
import sys

from openintel_amd import _lib

tag = os.environ.get("OI_LIB", "")
if tag.startswith("ablation"):
    assert _lib._lib is None, "tools/_ablation.py must be imported before the library is first used"
    _lib.LIB_PATH = os.path.join(_lib.HERE, "libopenintel_hip_%s.so" % tag)
    sys.stderr.write("[tools] loading the ABLATION build %s (timings only; OI_* switches live)\n" % _lib.LIB_PATH)
