"""Run an UNCHANGED reference script (e.g. its test.py) against the MI355X build:

    python -m multi_view_stereonet_amd.run_script /path/to/test.py <weights_dir> <data_dir> <test_file> ...

Registers mvsn::plane_sweep_forward (so that `torch.jit.load(weights_dir/"stereo_network.pt")`, test.py:311, resolves
the operator an archive from tools/make_archive.py refers to) and then executes the script as __main__."""
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    from . import torchscript  # noqa: F401  (operator registration)
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
