"""Importable alias for the package directory `binaural-audio-synthesis_amd/`.

The directory name contains hyphens, which the `import` statement cannot spell;
`import binaural_audio_synthesis_amd as bas` resolves to that package.
"""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("binaural-audio-synthesis_amd")
