"""The ctypes stub printed in INTEGRATION.md section 2 is executed as written (GPU only): documentation that
cannot rot."""
import os
import re

import numpy as np
import pytest

from parity_util import assert_obs_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_integration_md_stub_runs(golden):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# manytor_hip_binding\.py.*?)```", text, flags=re.S).group(1)
    code = code.replace("/root/repo/manytor_amd/libmanytor_hip.so", os.path.join(ROOT, "manytor_amd", "libmanytor_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = golden("f4_multienv_trace")
    me = ns["Multienv"](env_shape=(3, 2), obj_number=7)
    me.reset(g["points"][0])
    for t in range(3):
        obs2, reward, done = me.step([list(a) for a in g["action"][t]])
        assert len(obs2) == 6 and obs2[0].shape == (21,) and obs2[0].dtype == np.float64
        alive_before = g["alives"][t - 1] if t else np.ones((6, 7), dtype=bool)
        assert_obs_close(np.array(obs2), g["obs2"][t], g["jc"][t][:, 2], g["points"][0], alive_before)
        assert reward == [int(v) for v in g["reward"][t]]
        assert done == [bool(v) for v in g["done"][t]]
