"""The A/B switches that select kernel forms (DEMF_*, read once per process) must all give a correct step, not only
their defaults: VERDICT r5 "weak" 11 / ADVICE r5 found a combination that raised.  One mid-size training step per
non-default value of the most-used switches, each in its own process, against the default run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SETTINGS = [
    {},                                            # defaults (the reference of the comparison)
    {"DEMF_X3_MASK": "0"},                         # three-term launches -> the native fp32 MFMA kernels
    {"DEMF_X3_MASK": "5"},                         #   ... input-gradient launches only
    {"DEMF_F32_NATIVE": "1"},                      # the whole f32 mode on v_mfma_f32_32x32x2_f32
    {"DEMF_PERSIST_CUS": "256", "DEMF_PERSIST_CUS_BWD": "256"},
    {"DEMF_STATIC_TILES": "1"},                    # no dynamic tile counters, finalize as a separate launch
    {"DEMF_FWD_RES": "0", "DEMF_FWD_PC": "0"},     # the generic forward kernel everywhere
    {"DEMF_FWD_TILE": "0"},                        # few-row forward layers on mlp_gemm_kernel
    {"DEMF_NO_FIN": "1"},                          # BatchNorm bookkeeping as separate launches
    {"DEMF_DEFER_DW": "0"},                        # weight gradients launched where they arise
    {"DEMF_FPS_PAIR": "0", "DEMF_FPS_PRUNE": "0"},
    {"DEMF_GRAPH_UPDATE": "0", "_graph": "1"},     # captured step with the eager update behind it
    {"_graph": "1"},                               # captured step, update inside the graph
    {"DEMF_F16_TERMS": "0"},                       # three bf16 terms everywhere (no two-fp16-term forms)
    {"DEMF_F16_TERMS_BWD": "0"},                   #   ... in the fused backward kernels only
    {"DEMF_GF_RECOMPUTE": "0", "DEMF_GF_WACC": "0", "DEMF_GF_FIN": "0", "DEMF_ACC_REPL": "0"},   # round-6 forms off
    {"DEMF_INVERT_SPLIT": "2"},
]


def _run(extra):
    env = dict(os.environ)
    args = [sys.executable, os.path.join(ROOT, "tools", "switch_step.py")]
    for k, v in extra.items():
        if k == "_graph":
            args.append("--graph")
        else:
            env[k] = v
    out = subprocess.run(args, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (extra, out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_every_switch_value_gives_the_same_step():
    base = _run(SETTINGS[0])
    assert base["finite"] and base["grad_norm"] > 0 and base["param_delta"] > 0
    for extra in SETTINGS[1:]:
        got = _run(extra)
        assert got["finite"], extra
        # (the loss is taken AFTER an update: it inherits the gradients' sensitivity below.  2e-3 held while every form
        #  rounded its operands alike; the two-fp16-term kernels round at 2^-22 and scale the gradient operand per
        #  workgroup, so even the CU count moves the last bits: 3.6e-3 seen, 6e-3 allowed)
        assert got["loss"] == pytest.approx(base["loss"], rel=6e-3), (extra, got, base)
        # (untrained weights: near-tie flips move gradient norms by per cent between ANY two kernel forms - see
        # tests/parity_tools.py; a wrong kernel form moves them by tens of per cent or makes them non-finite)
        assert got["grad_norm"] == pytest.approx(base["grad_norm"], rel=8e-2), (extra, got, base)
        assert got["param_delta"] == pytest.approx(base["param_delta"], rel=8e-2), (extra, got, base)
