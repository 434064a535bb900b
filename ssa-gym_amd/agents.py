"""Device-side versions of the reference's heuristic agents (agents.py:7-81), same names and
`agent(obs, env)` signature.  The reference evaluates them with Python loops over all objects
(`visible_objects()` + list comprehensions over `env.P_filter[env.i]`), which at 20 000 objects costs
as much as the env step itself; here one kernel produces the visibility mask and
the per-object scores from the HBM-resident state and a second one does the masked arg-max, so an
agent call moves 16 bytes to the host.

Deviation: objects whose score is NaN (non positive definite covariance in the Shannon ratio) are
skipped by the arg-max; numpy's argmax would return the first NaN's index."""
import numpy as np


def _scores(env):
    """(scores[4, m], mask[m]) on the device for the env's current step."""
    return env.agent_scores()


def _pick(env, row, masked=True):
    from . import device
    scores, mask = _scores(env)
    j = device.masked_argmax(scores[row], mask if masked else None)
    return j if j >= 0 else env.action_space.sample()


def agent_naive_greedy(obs, env=None):          # agents.py:7  argmax trace(P)
    return _pick(env, 0, masked=False)


def agent_naive_random(obs=None, env=None):     # agents.py:12
    return env.action_space.sample()


def agent_shannon(obs, env):                    # agents.py:15  argmax log(det P_i / det P_{i-1}) over visible
    return _pick(env, 1)


def agent_visible_random(obs, env):             # agents.py:29
    visible = env.visible_objects()
    if not np.any(visible):
        return env.action_space.sample()
    return int(np.random.choice(visible))


def agent_visible_greedy(obs, env):             # agents.py:36  argmax trace(P) over visible
    return _pick(env, 0)


def agent_visible_greedy_spoiled(obs, env, p=0.25):   # agents.py:46
    greedy = _pick(env, 0)
    rand = env.action_space.sample()
    return int(np.random.choice(a=[greedy, rand], p=[1 - p, p]))


def agent_visible_greedy_aer(obs, env):         # agents.py:58  (trace P is the 4th aer-obs column)
    return _pick(env, 0)


def agent_pos_error_greedy(obs, env):           # agents.py:66
    return _pick(env, 2)


def agent_vel_error_greedy(obs, env):           # agents.py:75
    return _pick(env, 3)
