"""Host-side replay of the staging schedule of `encoder_wgrad_kernel` (mapf_rl_amd/csrc/mapf_wgrad.hip): the kernel keeps the
zero-bordered input images of its position stream in a 448-row circular LDS buffer that every 64-position block tops up
three blocks ahead.  The replay (tools/micro/wgrad_ring_check.py, same constants as the kernel) asserts that every row a
resident block reads through its 9 taps still holds exactly that row and that loads only ever touch interior rows."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("wgrad_ring_check", os.path.join(ROOT, "tools", "micro", "wgrad_ring_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ring_schedule_never_overwrites_a_needed_row():
    chk = _load()
    for nob in (1, 2, 3, 5, 6, 7, 8, 13, 29, 64):
        assert chk.check(nob) == (49 * nob + 63) // 64


def test_replay_uses_the_kernel_constants():
    chk = _load()
    src = open(os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_wgrad.hip")).read()
    const = {k: int(v) for k, v in re.findall(r"constexpr int (RING|CH_ROWS|WIN_CHUNKS) = (\d+)", src)}
    assert (const["RING"], const["CH_ROWS"], const["WIN_CHUNKS"]) == (chk.R, chk.CH_ROWS, chk.WIN)
