"""Zero-edit drop-in launcher: run the reference's own `train.py` / `generate_audio.py` on the MI355X hot path.

    python -m pix2pixhdaudiosr_amd.dropin /path/to/pix2pixHDAudioSR/train.py --name run1 --dataroot ... [reference flags]

The reference scripts import the hot path under top-level names (train.py:12-17,56-58; generate_audio.py:6-8,21-23):
`models.mdct`, `models.models`, `util.util`, `dct.dct`, `data.data_loader`.  A script's own directory always comes first on
`sys.path`, so alias packages beside this one could never shadow the reference's; instead this launcher installs ONE
meta-path finder in front of the import system that resolves exactly those names (and their siblings of the hot path) to
the modules of this package, and leaves everything §8 puts out of scope -- `options.*` (CLI), `util.visualizer`,
`util.html`, `util.spectro_img` (UI) -- to the reference's own files.  No line of the reference is edited.

The shipped reference hard-codes the MDCT2 transform (n_fft bins, models/pix2pixHD_model.py:37-40) and its eval code builds
the matching `IMDCT2(..., idct_op=IDCT())`; under this launcher the model therefore defaults to `mdct_type='mdct2'`
(P2PHD_MDCT_TYPE=mdct4 in the environment selects the n_fft/2-bin MDCT4 of BASELINE's 512x256 geometry instead).
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

PKG = __package__ or 'pix2pixhdaudiosr_amd'     # (`python -m` runs this file as __main__)

# reference name -> module of this package.  Whole packages: every submodule this build has is served; a submodule it does
# not have (the reference's deprecated / UI files) falls through to the reference's own directory.
ALIASES = {
    'models': PKG + '.models',
    'dct': PKG + '.dct',
    'data': PKG + '.data',
    'util.util': PKG + '.util.util',
}
# submodules of aliased packages that stay the reference's own (not part of the hot path)
PASS_THROUGH = ()


def _target(fullname):
    if fullname in ALIASES:
        return ALIASES[fullname]
    head, _, tail = fullname.partition('.')
    if head in ALIASES and tail and fullname not in PASS_THROUGH:
        return ALIASES[head] + '.' + tail
    return None


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)      # the SAME module object: `models.mdct.IMDCT2 is pkg.models.mdct.IMDCT2`

    def exec_module(self, module):
        pass


class AliasFinder(importlib.abc.MetaPathFinder):
    """Serves the reference's hot-path module names from this package."""

    def find_spec(self, fullname, path=None, target=None):
        t = _target(fullname)
        if t is None:
            return None
        try:
            found = importlib.util.find_spec(t)
        except (ImportError, ValueError):
            found = None
        if found is None:
            return None                                    # not a module of this build: the reference's own file, if any
        spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(t), is_package=found.submodule_search_locations is not None)
        if found.submodule_search_locations is not None:
            spec.submodule_search_locations = list(found.submodule_search_locations)
        return spec


def install():
    """Idempotent; returns the finder."""
    for f in sys.meta_path:
        if isinstance(f, AliasFinder):
            return f
    f = AliasFinder()
    sys.meta_path.insert(0, f)
    # a module imported BEFORE the hook under one of the names would win over it: refuse to run half-aliased
    clash = [n for n in list(sys.modules) if _target(n) is not None and getattr(sys.modules[n], '__name__', n) == n]
    for n in clash:
        del sys.modules[n]
    os.environ.setdefault('P2PHD_MDCT_TYPE', 'mdct2')
    return f


def uninstall():
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, AliasFinder)]
    for n in [n for n in sys.modules if _target(n) is not None]:
        del sys.modules[n]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 0 if argv else 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit("dropin: no such script: %s" % script)
    install()
    import runpy
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))             # what `python script.py` does: the script's directory first
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
