"""CPU: the early-stop drift diagnosis of the seeded fuzz (tests/test_gpu_parity.py::yardstick_instability) on scripted
trajectories -- the branch only ever runs on the GPU box, on the rare draw that misses its gate, so its three conditions are
pinned here with a scripted oracle and a scripted HIP path (fuzz 307/137, 309/232: profiles/r06_fuzz_307_309_trace.log):
excused only if (1) the two matched within the gate at an earlier step, (2) the oracle's cost on the deviating image did not
improve from that step to the batch-global stop, (3) the HIP path's final cost there is no higher than the oracle's."""
import numpy as np

from test_gpu_parity import FUZZ_GATE, yardstick_instability

N, B = 6, 2                      # the batch stops after N steps (early_stop: the trace ends there); image 1 is the one that drifts


def _state(k2, cost1):
    """A result dict: two radial images; image 0 pinned, image 1 at distortion k2 with final cost cost1."""
    cam = np.array([[64, 48, 50, 50, 32, 24, 0.01, 0.02], [64, 48, 60, 60, 32, 24, 0.05, k2]], np.float32)
    grav = np.array([[0, -1, 0], [0.1, -0.99, 0.05]], np.float32)
    return {"camera": cam, "gravity": grav, "final_cost": np.array([2e-4, cost1], np.float32),
            "initial_cost": np.array([1e-2, 1e-2], np.float32), "stop_at": np.full(B, N, np.float32)}


class ScriptedOracle:
    """oracle.solve(data, conf, precision=, trace=): the state after conf['num_steps'] steps from a script (both precisions alike)."""

    def __init__(self, k2_by_step, cost_by_step):
        self.k2, self.cost = k2_by_step, cost_by_step          # index = number of steps taken (0 .. N)

    def solve(self, data, conf, precision="f32", trace=False):
        k = N if conf.get("early_stop") else conf["num_steps"]
        out = _state(self.k2[k], self.cost[k])
        if trace:
            n = k
            cu = np.stack([np.array([2e-4, self.cost[i]], np.float32) for i in range(n)]) if n else np.zeros((0, B), np.float32)
            out["trace"] = {"lambda": np.ones((n, B), np.float32), "cost_up": cu, "cost_lat": np.zeros_like(cu),
                            "cam": np.stack([_state(self.k2[i + 1], 0)["camera"][:, 2:] for i in range(n)]),
                            "gravity": np.stack([_state(0, 0)["gravity"] for _ in range(n)])}
        return out


def _hip(k2_by_step, cost_by_step):
    def run(conf):
        k = N if conf.get("early_stop") else conf["num_steps"]
        return _state(k2_by_step[k], cost_by_step[k])
    return run


DATA = {"up_field": np.ones((B, 2, 4, 4), np.float32), "latitude_field": np.ones((B, 1, 4, 4), np.float32)}
CONF = {"camera_model": "radial", "num_steps": 20, "early_stop": True}
GATE = FUZZ_GATE["radial"]
# the oracle converges at step 3 and then creeps along k2 at a cost a hair higher; the HIP path stays where both converged
ORACLE_K2 = [0.5, 0.3, 0.12, 0.1000, 0.1000, 0.1003, 0.1006]
ORACLE_COST = [1e-2, 1e-3, 1.2e-4, 1.0e-4, 1.0e-4, 1.00001e-4, 1.00003e-4]
HIP_K2 = [0.5, 0.3, 0.12, 0.1000, 0.1000, 0.1000, 0.1000]
HIP_COST = [1e-2, 1e-3, 1.2e-4, 1.0e-4, 1.0e-4, 1.0e-4, 1.0e-4]
DEVIATION = np.array([0.0, 0.0, 6e-4, 3e-7])


def test_post_convergence_drift_before_a_batch_global_stop_is_recognised():
    why = yardstick_instability(ScriptedOracle(ORACLE_K2, ORACLE_COST), DATA, CONF, DEVIATION, GATE, hip=_hip(HIP_K2, HIP_COST))
    assert why is not None and "matched the oracle within the gate" in why and "[1]" in why and "after step 4" in why, why


def test_not_excused_while_the_oracle_was_still_descending():
    cost = list(ORACLE_COST)
    cost[4], cost[5] = 1.2e-4, 1.1e-4                  # the oracle's cost still fell on the way to the stop: no drift, a real difference
    why = yardstick_instability(ScriptedOracle(ORACLE_K2, cost), DATA, CONF, DEVIATION, GATE, hip=_hip(HIP_K2, HIP_COST))
    assert why is None, why


def test_not_excused_when_the_hip_path_ends_at_a_higher_cost():
    hip_cost = list(HIP_COST)
    hip_cost[N] = 1.01e-4                              # the HIP path's own final cost is the worse one: not the oracle's drift
    why = yardstick_instability(ScriptedOracle(ORACLE_K2, ORACLE_COST), DATA, CONF, DEVIATION, GATE, hip=_hip(HIP_K2, hip_cost))
    assert why is None, why


def test_not_excused_when_the_two_never_matched():
    hip_k2 = [0.5, 0.3, 0.12, 0.1010, 0.1010, 0.1010, 0.1010]          # 1e-3 apart at every step the oracle had converged
    why = yardstick_instability(ScriptedOracle(ORACLE_K2, ORACLE_COST), DATA, CONF, DEVIATION, GATE, hip=_hip(hip_k2, HIP_COST))
    assert why is None, why
