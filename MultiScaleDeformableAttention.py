"""Placeholder that is only imported when the COMPILED module of this name has not been built: the real
``MultiScaleDeformableAttention`` is the torch extension ``MultiScaleDeformableAttention.cpython-*.so`` next to this file
(openvis_amd/csrc/torch_ext/msda_module.cpp; an extension module takes precedence over a .py of the same name).  Like the
reference (ops/functions/ms_deform_attn_func.py:21-29) a missing build fails loudly, with the build hint."""
raise ModuleNotFoundError(
    "MultiScaleDeformableAttention has not been built.  Build it with:\n\n"
    "\t`python -c 'import __graft_entry__ as g; g.build()'`  (or `make -C openvis_amd/csrc torch_ext`)\n")
