"""Evaluation service (reference: cchess_alphazero/agent/api.py::CChessModelAPI).

The reference runs a prediction thread that drains multiprocessing pipes and calls Keras
``predict_on_batch``.  Here the network lives in the same process as the search kernels, so the
"pipe" is an object with two faces:
  * ``evaluate_device(planes_tensor) -> (policy, value)``  -- what the engine / CChessPlayer use
    (device tensors in, device tensors out, no copies);
  * ``send(list_of_planes)`` / ``poll()`` / ``recv()``  -- the reference's pipe protocol
    (player.py:108-143), for callers that still talk to it that way.
"""
import numpy as np
import torch

from cchess_alphazero.agent.model import InferenceNet


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
        if agent_model.model.cfg["cnn_filter_num"] not in (32, 128, 256):
            trunk = "library"
        self.net = InferenceNet(agent_model.model, dtype, trunk=trunk).to(self.device)
        self.done = False

    def start(self, need_reload=True):
        self.need_reload = need_reload

    def get_pipe(self, need_reload=True):
        pipe = DevicePipe(self)
        self.pipes.append(pipe)
        return pipe

    @torch.no_grad()
    def predict_device(self, planes):
        return self.net(planes)

    def close(self):
        self.done = True
