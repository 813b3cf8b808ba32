import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _fixture_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_cases():
    """Fixtures of the update path (MSCKF.update)."""
    return [n for n in _fixture_names() if not n.startswith(("sel_", "seq_", "assoc_"))]


def select_cases():
    """Fixtures of get_valid_features + the chained update."""
    return [n for n in _fixture_names() if n.startswith("sel_")]


def sequence_cases():
    """Fixtures of a multi-frame run: process_imu / state_augmentation / update / remove_cameras (f2, f3)."""
    return [n for n in _fixture_names() if n.startswith("seq_")]


def load_sequence(name):
    """Returns (header dict, list of per-op dicts with key 'kind')."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    kinds = z["op_kind"]
    ops = [dict(kind=int(k)) for k in kinds]
    head = {}
    for key in z.files:
        if key.startswith("o") and "_" in key and key[1:key.index("_")].isdigit():
            ops[int(key[1:key.index("_")])][key[key.index("_") + 1:]] = z[key]
        else:
            head[key] = z[key]
    return head, ops


def load_golden_select(name):
    """Returns (UpdateProblem, TrackTable, SelectParams, dict of expected outputs)."""
    from msckf_amd import synth
    prob, z = load_golden(name)
    tracks = synth.TrackTable(line_base=z["line_base"], line_dir=z["line_dir"], line_conf=z["line_conf"],
                              lost_for=z["lost_for"], tracked_for=z["tracked_for"])
    sp = z["select_params"]
    params = synth.SelectParams(min_frames_lost=int(sp[0]), min_frames_tracked=int(sp[1]), use_parallax=bool(sp[2]),
                                min_parallax_deg=float(sp[3]), width=int(sp[4]), height=int(sp[5]))
    return prob, tracks, params, z


def load_golden(name):
    """Returns (UpdateProblem, dict of expected outputs) for one fixture."""
    from msckf_amd import synth
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    prob = synth.UpdateProblem(
        P=z["P"], cam_R=z["cam_R"], cam_t=z["cam_t"], cam_R0=z["cam_R0"], cam_t0=z["cam_t0"],
        gravity=z["gravity"], K=z["K"], sigma=float(z["sigma"]), view_ptr=z["view_ptr"],
        obs_uv=z["obs_uv"], obs_slot=z["obs_slot"], idp_base=z["idp_base"], idp_m=z["idp_m"],
        idp_rho=z["idp_rho"], meta={"golden": name})
    return prob, {k: z[k] for k in z.files}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


@pytest.fixture(scope="session")
def engine_lib():
    """The HIP C-ABI library; GPU tests fail loudly if it is missing."""
    from msckf_amd import _ffi
    return _ffi.load()
