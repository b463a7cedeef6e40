"""Evaluation service (reference: cchess_alphazero/agent/api.py::CChessModelAPI).

The reference runs a prediction thread that drains multiprocessing pipes and calls Keras
``predict_on_batch``.  Here the network lives in the same process as the search kernels, so the
"pipe" is an object with two faces:
  * ``evaluate_device(planes_tensor) -> (policy, value)``  -- what the engine / CChessPlayer use
    (device tensors in, device tensors out, no copies);
  * ``send(list_of_planes)`` / ``poll()`` / ``recv()``  -- the reference's pipe protocol
    (player.py:108-143), for callers that still talk to it that way.
"""
from logging import getLogger
from time import time

import os
import numpy as np
import torch

from cchess_alphazero.agent.model import guarded_inference_net


logger = getLogger(__name__)


class DevicePipe:
    def __init__(self, api):
        self.api = api
        self._replies = []

    def evaluate_device(self, planes):
        return self.api.predict_device(planes)

    # -- reference pipe protocol --
    def send(self, data):
        planes = torch.from_numpy(np.asarray(data, dtype=np.float32)).to(self.api.device)
        p, v = self.api.predict_device(planes)
        p, v = p.cpu().numpy(), v.cpu().numpy()
        self._replies.append([(p[i], float(v[i])) for i in range(len(data))])

    def poll(self, timeout=None):
        return bool(self._replies)

    def recv(self):
        return self._replies.pop(0)

    def close(self):
        pass


class CChessModelAPI:
    def __init__(self, config, agent_model, dtype=None, device=None):
        self.config = config
        self.agent_model = agent_model
        self.pipes = []
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        dtype = dtype or getattr(torch, getattr(getattr(config, "engine", None), "net_dtype", "float32"))
        trunk = getattr(getattr(config, "engine", None), "net_trunk", "mfma")
        self._arith = os.environ.get("CZ_TOWER_ARITH") or getattr(getattr(config, "engine", None), "net_arith", "bf16x3")
        if agent_model.model.cfg["cnn_filter_num"] not in (32, 128, 192, 256):
            trunk = "library"
        self._guard = None if getattr(getattr(config, "engine", None), "arith_guard", True) else False
        # (the tower arithmetic is a request: it is measured against the float64 network on calibration positions and
        #  replaced by a more exact one when these weights need it -- agent/model.py guarded_inference_net)
        self.net = guarded_inference_net(agent_model.model, dtype, trunk=trunk, arith=self._arith, device=self.device,
                                         guard=self._guard)
        self._dtype, self._trunk = dtype, trunk
        self.done = False
        self.need_reload = True
        self.reload_interval = 600.0             # seconds between checks of the best-weight file (api.py:42-44)
        self._last_check = time()

    def start(self, need_reload=True):
        self.need_reload = need_reload

    def try_reload_model(self):
        """Hot reload (api.py:76-87): when the best-weight file on disk has a different digest, load it and rebuild
        the inference network.  Returns True when new weights were loaded."""
        from cchess_alphazero.lib.model_helper import load_best_model_weight, need_to_reload_best_model_weight
        try:
            if self.need_reload and need_to_reload_best_model_weight(self.agent_model):
                if load_best_model_weight(self.agent_model):
                    trunk = self._trunk
                    if self.agent_model.model.cfg["cnn_filter_num"] not in (32, 128, 192, 256):
                        trunk = "library"
                    self.net = guarded_inference_net(self.agent_model.model, self._dtype, trunk=trunk, arith=self._arith,
                                                     device=self.device, guard=self._guard)
                    return True
        except Exception as e:                    # a half-written file: keep serving the old weights
            logger.error(e)
        return False

    def get_pipe(self, need_reload=True):
        pipe = DevicePipe(self)
        self.pipes.append(pipe)
        return pipe

    @torch.no_grad()
    def predict_device(self, planes):
        if self.need_reload and time() - self._last_check > self.reload_interval:
            self._last_check = time()
            self.try_reload_model()
        return self.net(planes)

    def close(self):
        self.done = True
