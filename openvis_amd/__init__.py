"""openvis_amd — MI355X-native per-frame dense inference path of OpenVIS.

Host-side mirror (Python, like the reference) of the reference's operator/module
interface for the hot path, over the C-ABI HIP library ``lib/libopenvis_hip.so``
(sources in ``csrc/``, declarations in ``/include/openvis_hip.h``).
There is no CPU or eager fallback: every op raises if the HIP library is missing.
"""
__version__ = "0.1.0"
