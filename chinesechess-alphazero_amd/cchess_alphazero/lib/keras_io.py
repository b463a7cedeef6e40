"""Keras model files of the reference -> the torch network.

The reference stores a model as ``Model.get_config()`` JSON (data/model/*.json) plus ``save_weights`` HDF5
(cchess_alphazero/agent/model.py:95-115).  This module reads both without Keras / TensorFlow / h5py:

    cfg = config_from_keras(json.load(open(".../model_best_config.json")))   # CChessNet(**cfg)
    net = CChessNet(**cfg); load_keras_weights(net, ".../model_best_weight.h5", names_from_keras(json_cfg))

Layout conversions (Keras 2.0.8, data_format channels_first; agent/model.py:36-83):
    Conv2D kernel  [kh, kw, in, out]  -> torch [out, in, kh, kw]
    Dense kernel   [in, out]          -> torch [out, in]           (Flatten order C,H,W on both sides)
    BatchNorm      gamma, beta, moving_mean, moving_variance -> weight, bias, running_mean, running_var
"""
import numpy as np

from cchess_alphazero.lib import hdf5_min


class KerasFormatError(ValueError):
    pass


def _layers(keras_cfg):
    layers = keras_cfg.get("layers") or keras_cfg.get("config", {}).get("layers")
    if not layers:
        raise KerasFormatError("not a Keras functional-model config (no 'layers')")
    return layers


def names_from_keras(keras_cfg):
    """Role -> Keras layer name, following the graph of CChessModel.build (agent/model.py:36-83)."""
    by_name = {l["name"]: l for l in _layers(keras_cfg)}

    def inbound(layer):
        nodes = layer.get("inbound_nodes") or []
        return [x[0] for x in nodes[0]] if nodes else []

    def consumers(name):
        return [l for l in _layers(keras_cfg) if name in inbound(l)]

    def only(ls, what):
        if len(ls) != 1:
            raise KerasFormatError(f"expected exactly one {what}, found {[l['name'] for l in ls]}")
        return ls[0]

    def follow(name, cls):
        return only([l for l in consumers(name) if l["class_name"] == cls], f"{cls} after {name}")

    inp = only([l for l in _layers(keras_cfg) if l["class_name"] == "InputLayer"], "InputLayer")
    conv = follow(inp["name"], "Conv2D")
    bn = follow(conv["name"], "BatchNormalization")
    act = follow(bn["name"], "Activation")
    names = {"input": inp["name"], "input_conv": conv["name"], "input_bn": bn["name"], "res": []}
    x = act["name"]
    while True:
        convs = [l for l in consumers(x) if l["class_name"] == "Conv2D"]
        adds = [l for l in consumers(x) if l["class_name"] == "Add"]
        if len(convs) == 1 and len(adds) == 1:           # a residual block: conv-bn-relu-conv-bn-add-relu
            c1 = convs[0]
            b1 = follow(c1["name"], "BatchNormalization")
            r1 = follow(b1["name"], "Activation")
            c2 = follow(r1["name"], "Conv2D")
            b2 = follow(c2["name"], "BatchNormalization")
            add = adds[0]
            if b2["name"] not in inbound(add):
                raise KerasFormatError(f"residual block at {x}: {add['name']} does not add {b2['name']}")
            names["res"].append((c1["name"], b1["name"], c2["name"], b2["name"]))
            x = follow(add["name"], "Activation")["name"]
            continue
        break
    heads = [l for l in consumers(x) if l["class_name"] == "Conv2D"]
    if len(heads) != 2:
        raise KerasFormatError(f"expected the policy and value 1x1 convolutions after {x}")
    for h in heads:
        hb = follow(h["name"], "BatchNormalization")
        fl = follow(follow(hb["name"], "Activation")["name"], "Flatten")
        dense = follow(fl["name"], "Dense")
        if dense["config"]["activation"] == "softmax":
            names.update(policy_conv=h["name"], policy_bn=hb["name"], policy_out=dense["name"])
        else:
            out = follow(dense["name"], "Dense")
            names.update(value_conv=h["name"], value_bn=hb["name"], value_dense=dense["name"], value_out=out["name"])
    if "policy_out" not in names or "value_out" not in names:
        raise KerasFormatError("could not identify the policy / value heads")
    return names


def config_from_keras(keras_cfg):
    """Keras JSON -> keyword arguments of cchess_alphazero.agent.model.CChessNet."""
    by_name = {l["name"]: l["config"] for l in _layers(keras_cfg)}
    n = names_from_keras(keras_cfg)
    first = by_name[n["input_conv"]]
    if first.get("data_format", "channels_first") != "channels_first":
        raise KerasFormatError("only channels_first models are supported")
    block_conv = by_name[n["res"][0][0]] if n["res"] else first
    cfg = dict(cnn_filter_num=int(first["filters"]), cnn_first_filter_size=int(first["kernel_size"][0]),
               cnn_filter_size=int(block_conv["kernel_size"][0]), res_layer_num=len(n["res"]),
               value_fc_size=int(by_name[n["value_dense"]]["units"]),
               input_depth=int(by_name[n["input"]]["batch_input_shape"][1]),
               policy_filters=int(by_name[n["policy_conv"]]["filters"]),
               value_filters=int(by_name[n["value_conv"]]["filters"]),
               n_labels=int(by_name[n["policy_out"]]["units"]))
    return cfg


def _layer_weights(f, layer):
    if layer not in f:
        raise KerasFormatError(f"weight file has no group for layer {layer!r}")
    g = f[layer]
    names = g.attrs.get("weight_names")
    if names is None:
        raise KerasFormatError(f"layer {layer!r}: no weight_names attribute")
    out = {}
    for wn in np.atleast_1d(names):
        wn = wn.decode("utf8") if isinstance(wn, bytes) else str(wn)
        out[wn.split("/")[-1].split(":")[0]] = g[wn].read()
    return out


def load_keras_weights(net, h5_path, names):
    """Copy the weights of a Keras ``save_weights`` file into a CChessNet (shapes are checked)."""
    import torch

    def put(param, arr, what):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        if tuple(t.shape) != tuple(param.shape):
            raise KerasFormatError(f"{what}: file has shape {tuple(t.shape)}, the network expects {tuple(param.shape)}")
        with torch.no_grad():
            param.copy_(t)

    with hdf5_min.File(h5_path) as f:
        def conv(mod, layer):
            put(mod.weight, _layer_weights(f, layer)["kernel"].transpose(3, 2, 0, 1), layer + "/kernel")

        def bn(mod, layer):
            w = _layer_weights(f, layer)
            put(mod.weight, w["gamma"], layer + "/gamma")
            put(mod.bias, w["beta"], layer + "/beta")
            put(mod.running_mean, w["moving_mean"], layer + "/moving_mean")
            put(mod.running_var, w["moving_variance"], layer + "/moving_variance")

        def dense(mod, layer):
            w = _layer_weights(f, layer)
            put(mod.weight, w["kernel"].T, layer + "/kernel")
            put(mod.bias, w["bias"], layer + "/bias")

        conv(net.input_conv, names["input_conv"])
        bn(net.input_bn, names["input_bn"])
        if len(names["res"]) != len(net.res):
            raise KerasFormatError("residual block count of the config and the network differ")
        for blk, (c1, b1, c2, b2) in zip(net.res, names["res"]):
            conv(blk.conv1, c1); bn(blk.bn1, b1); conv(blk.conv2, c2); bn(blk.bn2, b2)
        conv(net.policy_conv, names["policy_conv"]); bn(net.policy_bn, names["policy_bn"])
        dense(net.policy_out, names["policy_out"])
        conv(net.value_conv, names["value_conv"]); bn(net.value_bn, names["value_bn"])
        dense(net.value_dense, names["value_dense"]); dense(net.value_out, names["value_out"])
    return net
