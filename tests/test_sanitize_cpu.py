"""CPU sanitizer target (round-4 review, SURVEY section 5 aux): every host-only piece of native code -- the CBS / space-time A* planner
(csrc/mapf_search.hip, reference search.py:58-442), the scenario generator behind mapf_generate (csrc/mapf_generate_host.inc,
reference environment.py:21-138) and the CPU oracle (oracle/mapf_oracle.c) -- built with g++ / gcc -fsanitize=address,undefined
(the .hip source with -x c++; nothing here touches a GPU) and run over the search goldens, a generator sweep and oracle rollouts.
The instrumented run must be clean (no report, exit code 0) and must produce what the product library produces."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "sanitize")
SWEEP = [(16, 10, 1, -1.0, 1), (16, 20, 6, -1.0, 2), (8, 32, 40, 0.3, 3), (4, 40, 16, 0.3, 4), (2, 64, 128, 0.3, 5), (8, 16, 40, 0.3, 6),
         (32, 8, 4, 0.45, 7), (4, 12, 60, 0.1, 8), (3, 5, 3, 0.0, 9)]
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def driver():
    os.makedirs(OUT, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mapf_rl_amd", "csrc")]
    objs = []
    for src, cc, lang in ((os.path.join(ROOT, "mapf_rl_amd", "csrc", "mapf_search.hip"), "g++", ["-x", "c++", "-std=c++17"]),
                          (os.path.join(ROOT, "tests", "sanitize", "host_driver.cc"), "g++", ["-std=c++17"]),
                          (os.path.join(ROOT, "oracle", "mapf_oracle.c"), "gcc", ["-std=c11"])):
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        subprocess.check_call([cc] + lang + SAN + ["-Wall"] + inc + ["-c", src, "-o", obj])
        objs.append(obj)
    exe = os.path.join(OUT, "host_driver")
    subprocess.check_call(["g++"] + SAN + objs + ["-o", exe])
    return exe


def test_host_code_is_clean_under_address_and_undefined_behaviour_sanitizers(driver, tmp_path):
    z = H.load_npz("search.npz")
    n = int(z["num_cases"])
    cases = []
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<i", n))
        for k in range(n):
            pre = "case%d_" % k
            m = np.ascontiguousarray(z[pre + "map"] != 0, dtype=np.int8)
            a, g = np.ascontiguousarray(z[pre + "agents"], dtype=np.int16), np.ascontiguousarray(z[pre + "goals"], dtype=np.int16)
            f.write(struct.pack("<ii", m.shape[0], a.shape[0]))
            f.write(m.tobytes() + a.tobytes() + g.tobytes())
            cases.append((m, a, g, int(z[pre + "ref_cost"]), z[pre + "dist0"]))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "240"], capture_output=True, text=True, env=env, timeout=1500)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr and "LeakSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "host_driver ok: %d search cases" % n in run.stdout
    buf = open(tmp_path / "out.bin", "rb").read()
    off = 0
    # ---- planner: every case solved within the instrumented build's time limit, plans valid and never costlier than the reference's ----
    for m, a, g, ref_cost, dist0 in cases:
        st, steps, cost = struct.unpack_from("<iii", buf, off)
        off += 12
        N, L = a.shape[0], m.shape[0]
        acts = np.frombuffer(buf, np.int8, steps * N, off).reshape(steps, N)
        off += steps * N
        dist = np.frombuffer(buf, np.int32, L * L, off).reshape(L, L)
        off += 4 * L * L
        assert st == 0 and cost <= ref_cost, (st, cost, ref_cost)
        ag, done = a.copy(), False
        for t in range(steps):
            s, ag, rc, done = oracle.step(m, ag, g, acts[t])
            assert s == 0 and not np.any(rc == oracle.RC_COLLISION)
        assert done or steps == 0
        assert np.array_equal(dist, dist0)
    # ---- generator: byte for byte what the product library's mapf_generate returns ----
    import mapf_rl_amd as M

    for E, L, N, rho, seed in SWEEP:
        st, redraws = struct.unpack_from("<ii", buf, off)
        off += 8
        maps = np.frombuffer(buf, np.int8, E * L * L, off).reshape(E, L, L)
        off += E * L * L
        ag = np.frombuffer(buf, np.int16, E * N * 2, off).reshape(E, N, 2)
        off += E * N * 4
        gl = np.frombuffer(buf, np.int16, E * N * 2, off).reshape(E, N, 2)
        off += E * N * 4
        assert st == 0, (E, L, N, rho, st)
        pm, pa, pg, pr = M.generate_scenarios(E, L, N, rho, seed=seed)
        assert np.array_equal(maps, pm) and np.array_equal(ag, pa) and np.array_equal(gl, pg) and redraws == pr, (E, L, N, rho)
    assert off + 8 == len(buf)
