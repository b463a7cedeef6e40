#!/opt/conda/bin/python3.9
"""Generates the HDF5 fixtures of tests/test_keras_io.py with the REAL HDF5 library (h5py 3.3 in /opt/conda's
python3.9 -- the main interpreter of this image has no h5py), so that the pure-Python reader
(cchess_alphazero/lib/hdf5_min.py) is checked against files libhdf5 wrote:

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py

  hdf5_misc.h5        nested groups (enough members to split symbol nodes), datasets and attributes of every supported
                      kind, many attributes on one object (continuation blocks)
  hdf5_chunked.h5 / hdf5_latest.h5   a chunked + gzip dataset and a libver="latest" file: must be refused, not mis-read
  keras_tiny.json     Model.get_config()-style topology of a 2-block x 32-filter network with the layer names, graph
                      and per-layer config keys of the reference's data/model/model_128f.json
  keras_tiny.h5       its weights, written exactly the way Keras 2.0.8 save_weights does (topology.py
                      save_weights_to_hdf5_group: root attrs layer_names / backend / keras_version, one group per layer
                      with attr weight_names, datasets named '<layer>/<weight>:0'); values from
                      numpy.random.RandomState(20260923) in file order (the test regenerates them).
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def misc():
    with h5py.File(os.path.join(HERE, "hdf5_misc.h5"), "w") as f:
        f.attrs["title"] = np.bytes_("misc fixture")
        f.attrs["numbers"] = np.arange(5, dtype=np.int32)
        f.attrs["pi"] = np.float64(3.141592653589793)
        f.attrs["empty"] = np.zeros((0,), dtype=np.float64)
        f.attrs["names"] = np.array([b"alpha", b"be", b"gamma-delta"])
        f.attrs["vlen_str"] = ["x", "yy", "third one"]            # h5py 3 stores these as variable-length strings
        f.attrs["vlen_bytes"] = [b"p", b"qq"]
        f.attrs["vlen_scalar"] = "just one"
        g = f.create_group("grp")
        for i in range(40):                                   # > 2 * leaf K: several symbol nodes under a B-tree
            g.create_dataset(f"d{i:02d}", data=np.full((3,), i, dtype=np.float32))
        sub = g.create_group("sub/deeper")
        sub.create_dataset("m", data=np.arange(24, dtype=np.float64).reshape(2, 3, 4))
        sub.create_dataset("i16", data=np.array([-3, 7, 11], dtype=np.int16))
        sub.create_dataset("scalar", data=np.float32(2.5))
        sub.create_dataset("u8", data=np.arange(7, dtype=np.uint8))
        many = f.create_group("many_attrs")
        for i in range(30):                                   # object header continuation blocks
            many.attrs[f"a{i:02d}"] = np.arange(i + 1, dtype=np.float32)
        f.create_dataset("with/slash:0", data=np.linspace(0, 1, 11, dtype=np.float32))


def keras_tiny():
    F, BLOCKS, VFC, LABELS, PF, VF = 32, 2, 16, 50, 2, 4
    layers = []

    def add(cls, name, cfg, inbound):
        layers.append({"name": name, "class_name": cls, "config": dict(name=name, **cfg),
                       "inbound_nodes": [[[i, 0, 0, {}] for i in inbound]] if inbound else []})

    conv = lambda f, k: dict(trainable=True, filters=f, kernel_size=[k, k], strides=[1, 1], padding="same",
                             data_format="channels_first", dilation_rate=[1, 1], activation="linear", use_bias=False)
    bn = dict(trainable=True, axis=1, momentum=0.99, epsilon=0.001, center=True, scale=True)
    add("InputLayer", "input_1", dict(batch_input_shape=[None, 14, 10, 9], dtype="float32", sparse=False), [])
    add("Conv2D", f"input_conv-5-{F}", conv(F, 5), ["input_1"])
    add("BatchNormalization", "input_batchnorm", bn, [f"input_conv-5-{F}"])
    add("Activation", "input_relu", dict(trainable=True, activation="relu"), ["input_batchnorm"])
    x = "input_relu"
    for i in range(1, BLOCKS + 1):
        add("Conv2D", f"res{i}_conv1-3-{F}", conv(F, 3), [x])
        add("BatchNormalization", f"res{i}_batchnorm1", bn, [f"res{i}_conv1-3-{F}"])
        add("Activation", f"res{i}_relu1", dict(trainable=True, activation="relu"), [f"res{i}_batchnorm1"])
        add("Conv2D", f"res{i}_conv2-3-{F}", conv(F, 3), [f"res{i}_relu1"])
        add("BatchNormalization", f"res{i}_batchnorm2", bn, [f"res{i}_conv2-3-{F}"])
        add("Add", f"res{i}_add", dict(trainable=True), [x, f"res{i}_batchnorm2"])
        add("Activation", f"res{i}_relu2", dict(trainable=True, activation="relu"), [f"res{i}_add"])
        x = f"res{i}_relu2"
    add("Conv2D", f"value_conv-1-{VF}", conv(VF, 1), [x])
    add("Conv2D", f"policy_conv-1-{PF}", conv(PF, 1), [x])
    add("BatchNormalization", "value_batchnorm", bn, [f"value_conv-1-{VF}"])
    add("BatchNormalization", "policy_batchnorm", bn, [f"policy_conv-1-{PF}"])
    add("Activation", "value_relu", dict(trainable=True, activation="relu"), ["value_batchnorm"])
    add("Activation", "policy_relu", dict(trainable=True, activation="relu"), ["policy_batchnorm"])
    add("Flatten", "value_flatten", dict(trainable=True), ["value_relu"])
    add("Flatten", "policy_flatten", dict(trainable=True), ["policy_relu"])
    add("Dense", "value_dense", dict(trainable=True, units=VFC, activation="relu", use_bias=True), ["value_flatten"])
    add("Dense", "policy_out", dict(trainable=True, units=LABELS, activation="softmax", use_bias=True), ["policy_flatten"])
    add("Dense", "value_out", dict(trainable=True, units=1, activation="tanh", use_bias=True), ["value_dense"])
    cfg = {"name": "cchess_model", "layers": layers, "input_layers": [["input_1", 0, 0]],
           "output_layers": [["policy_out", 0, 0], ["value_out", 0, 0]]}
    with open(os.path.join(HERE, "keras_tiny.json"), "w") as f:
        json.dump(cfg, f)

    rng = np.random.RandomState(20260923)
    by_name = {l["name"]: l for l in layers}

    def in_channels(layer):
        src = by_name[layer["inbound_nodes"][0][0][0]]
        while src["class_name"] not in ("Conv2D", "InputLayer"):
            src = by_name[src["inbound_nodes"][0][0][0]]
        return 14 if src["class_name"] == "InputLayer" else src["config"]["filters"]

    def in_features(layer):
        src = by_name[layer["inbound_nodes"][0][0][0]]
        if src["class_name"] == "Dense":
            return src["config"]["units"]
        while src["class_name"] != "Conv2D":
            src = by_name[src["inbound_nodes"][0][0][0]]
        return src["config"]["filters"] * 90

    with h5py.File(os.path.join(HERE, "keras_tiny.h5"), "w") as f:
        # h5py 2.x (the Keras 2.0.8 era) stored lists of bytes as fixed-length strings; h5py 3 needs the dtype spelled out
        f.attrs["layer_names"] = np.array([l["name"].encode("utf8") for l in layers], dtype="S")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["keras_version"] = "2.0.8".encode("utf8")
        for l in layers:
            g = f.create_group(l["name"])
            c = l["config"]
            if l["class_name"] == "Conv2D":
                k = c["kernel_size"][0]
                ws = [("kernel:0", rng.randn(k, k, in_channels(l), c["filters"]) * 0.1)]
            elif l["class_name"] == "BatchNormalization":
                n = by_name[l["inbound_nodes"][0][0][0]]["config"]["filters"]
                ws = [("gamma:0", 1 + 0.2 * rng.randn(n)), ("beta:0", 0.2 * rng.randn(n)),
                      ("moving_mean:0", 0.3 * rng.randn(n)), ("moving_variance:0", rng.uniform(0.5, 2.0, n))]
            elif l["class_name"] == "Dense":
                ws = [("kernel:0", rng.randn(in_features(l), c["units"]) * 0.05), ("bias:0", 0.1 * rng.randn(c["units"]))]
            else:
                ws = []
            names = [(l["name"] + "/" + n).encode("utf8") for n, _ in ws]
            g.attrs["weight_names"] = np.array(names, dtype="S") if names else []
            for name, (_, val) in zip(names, ws):
                val = val.astype(np.float32)
                d = g.create_dataset(name, val.shape, dtype=val.dtype)
                d[...] = val


def unsupported():
    """files the reader must refuse with a clear error instead of mis-reading"""
    with h5py.File(os.path.join(HERE, "hdf5_chunked.h5"), "w") as f:
        f.create_dataset("c", data=np.arange(100, dtype=np.float32).reshape(10, 10), chunks=(5, 5), compression="gzip")
        f.create_dataset("plain", data=np.arange(4, dtype=np.float32))
    with h5py.File(os.path.join(HERE, "hdf5_latest.h5"), "w", libver="latest") as f:
        f.create_dataset("x", data=np.arange(4, dtype=np.float32))


if __name__ == "__main__":
    misc()
    keras_tiny()
    unsupported()
    for n in ("hdf5_misc.h5", "keras_tiny.json", "keras_tiny.h5", "hdf5_chunked.h5", "hdf5_latest.h5"):
        print(n, os.path.getsize(os.path.join(HERE, n)))
