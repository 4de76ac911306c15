"""Import alias: ``import iq_to_audio_amd`` loads the package that lives in the directory
``iq-to-audio_amd/`` (a hyphen is not legal in a Python module name)."""
from __future__ import annotations

import pathlib as _pathlib

_real = _pathlib.Path(__file__).resolve().parent.parent / "iq-to-audio_amd"
__path__ = [str(_real)]
__file__ = str(_real / "__init__.py")
exec(compile((_real / "__init__.py").read_text(), __file__, "exec"))
