"""abyss_amd: the abyss-bloom-dbg unitig stage (Bloom-filter de Bruijn graph) on MI355X.

Package contents: ``csrc/`` (HIP kernels + C ABI, built into ``lib/libabyss_amd.so``),
``api.py`` (ctypes mirror of the reference's assembly interface), ``synth.py`` (seeded
synthetic read sets), ``build.py`` (in-tree build helpers).
"""
__all__ = ["api", "build", "synth"]
