"""CPU tests of the reference-model import: the pure-Python HDF5 reader against files written by the real
HDF5 library (fixtures from tests/golden/make_keras_h5.py, h5py 3.3), the Keras JSON -> CChessNet mapping on every
topology file the reference ships, and a forward pass of the imported network against a NumPy evaluation of the same
Keras graph."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def test_hdf5_reader_on_library_written_file():
    from cchess_alphazero.lib import hdf5_min
    with hdf5_min.File(os.path.join(GOLD, "hdf5_misc.h5")) as f:
        assert f.attrs["title"] == b"misc fixture"
        assert f.attrs["numbers"].tolist() == [0, 1, 2, 3, 4] and f.attrs["numbers"].dtype == np.int32
        assert f.attrs["pi"] == 3.141592653589793
        assert f.attrs["empty"].shape == (0,)
        assert [x.decode() for x in f.attrs["names"]] == ["alpha", "be", "gamma-delta"]
        assert [x.decode() for x in f.attrs["vlen_str"]] == ["x", "yy", "third one"]
        assert list(f.attrs["vlen_bytes"]) == [b"p", b"qq"] and f.attrs["vlen_scalar"] == b"just one"
        g = f["grp"]
        assert g.keys() == [f"d{i:02d}" for i in range(40)] + ["sub"]
        for i in (0, 17, 39):
            assert g[f"d{i:02d}"].read().tolist() == [float(i)] * 3
        sub = f["grp/sub/deeper"]
        assert np.array_equal(sub["m"].read(), np.arange(24, dtype=np.float64).reshape(2, 3, 4))
        assert sub["i16"].read().tolist() == [-3, 7, 11] and sub["i16"].read().dtype == np.int16
        assert sub["scalar"].read() == np.float32(2.5) and sub["scalar"].shape == ()
        assert sub["u8"].read().tolist() == list(range(7))
        many = f["many_attrs"]
        assert len(many.attrs) == 30
        for i in (0, 13, 29):
            assert np.array_equal(many.attrs[f"a{i:02d}"], np.arange(i + 1, dtype=np.float32))
        assert np.allclose(f["with/slash:0"].read(), np.linspace(0, 1, 11, dtype=np.float32))
        assert "grp/nope" not in f and "grp/d05" in f
        with pytest.raises(KeyError):
            f["grp/nope"]


def test_hdf5_reader_rejects_other_files(tmp_path):
    from cchess_alphazero.lib import hdf5_min
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(hdf5_min.Hdf5Error):
        hdf5_min.File(str(p))
    # real HDF5 the reader does not implement: a clear error, never a wrong array
    with hdf5_min.File(os.path.join(GOLD, "hdf5_chunked.h5")) as f:
        assert f["plain"].read().tolist() == [0.0, 1.0, 2.0, 3.0]
        with pytest.raises(hdf5_min.Hdf5Error, match="chunked"):
            f["c"].read()
    with pytest.raises(hdf5_min.Hdf5Error, match="superblock"):
        hdf5_min.File(os.path.join(GOLD, "hdf5_latest.h5"))


@pytest.mark.parametrize("name,want", [
    ("model_128f", dict(cnn_filter_num=128, res_layer_num=7, input_depth=14, policy_filters=2, value_filters=4)),
    ("model_256f", dict(cnn_filter_num=256, res_layer_num=7, input_depth=14, policy_filters=2, value_filters=4)),
    ("model_192x10_config", dict(cnn_filter_num=192, res_layer_num=10, input_depth=14, policy_filters=4, value_filters=2)),
    ("model_128_l1_config", dict(cnn_filter_num=128, res_layer_num=7, input_depth=28, policy_filters=32, value_filters=4)),
])
def test_reference_topologies(name, want):
    """The facts below were read off the reference's own data/model/*.json with the importer (golden: the values are
    committed here because /root/reference does not travel); when the reference is present the files are re-parsed."""
    from cchess_alphazero.lib import keras_io
    path = f"/root/reference/data/model/{name}.json"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present on this machine")
    cfg = keras_io.config_from_keras(json.load(open(path)))
    for k, v in want.items():
        assert cfg[k] == v
    assert cfg["cnn_first_filter_size"] == 5 and cfg["cnn_filter_size"] == 3 and cfg["value_fc_size"] == 256
    assert cfg["n_labels"] == 2086


def _tiny_weights():
    """the arrays make_keras_h5.py wrote, regenerated from the same RandomState stream"""
    cfg = json.load(open(os.path.join(GOLD, "keras_tiny.json")))
    rng = np.random.RandomState(20260923)
    by = {l["name"]: l for l in cfg["layers"]}
    out = {}

    def src_conv(l):
        s = by[l["inbound_nodes"][0][0][0]]
        while s["class_name"] not in ("Conv2D", "InputLayer"):
            s = by[s["inbound_nodes"][0][0][0]]
        return s

    for l in cfg["layers"]:
        c = l["config"]
        if l["class_name"] == "Conv2D":
            s = src_conv(l)
            cin = 14 if s["class_name"] == "InputLayer" else s["config"]["filters"]
            k = c["kernel_size"][0]
            out[l["name"]] = dict(kernel=(rng.randn(k, k, cin, c["filters"]) * 0.1).astype(np.float32))
        elif l["class_name"] == "BatchNormalization":
            n = by[l["inbound_nodes"][0][0][0]]["config"]["filters"]
            out[l["name"]] = dict(gamma=(1 + 0.2 * rng.randn(n)).astype(np.float32), beta=(0.2 * rng.randn(n)).astype(np.float32),
                                  moving_mean=(0.3 * rng.randn(n)).astype(np.float32),
                                  moving_variance=rng.uniform(0.5, 2.0, n).astype(np.float32))
        elif l["class_name"] == "Dense":
            s = by[l["inbound_nodes"][0][0][0]]
            fin = s["config"]["units"] if s["class_name"] == "Dense" else src_conv(l)["config"]["filters"] * 90
            out[l["name"]] = dict(kernel=(rng.randn(fin, c["units"]) * 0.05).astype(np.float32),
                                  bias=(0.1 * rng.randn(c["units"])).astype(np.float32))
    return cfg, out


def _keras_forward_numpy(cfg, w, x):
    """Evaluates the Keras graph layer by layer in NumPy float64 (channels_first, 'same' padding, BN in inference mode)."""
    acts = {}
    for l in cfg["layers"]:
        name, c = l["name"], l["config"]
        ins = [acts[i[0]] for i in l["inbound_nodes"][0]] if l["inbound_nodes"] else []
        if l["class_name"] == "InputLayer":
            acts[name] = x.astype(np.float64)
        elif l["class_name"] == "Conv2D":
            k = w[name]["kernel"].astype(np.float64)
            kh = k.shape[0]
            pad = kh // 2
            xi = np.pad(ins[0], ((0, 0), (0, 0), (pad, pad), (pad, pad)))
            y = np.zeros((xi.shape[0], k.shape[3], 10, 9))
            for dy in range(kh):
                for dx in range(kh):
                    y += np.einsum("nchw,co->nohw", xi[:, :, dy:dy + 10, dx:dx + 9], k[dy, dx])
            acts[name] = y
        elif l["class_name"] == "BatchNormalization":
            p = {k: v.astype(np.float64)[None, :, None, None] for k, v in w[name].items()}
            acts[name] = (ins[0] - p["moving_mean"]) / np.sqrt(p["moving_variance"] + c["epsilon"]) * p["gamma"] + p["beta"]
        elif l["class_name"] == "Activation":
            acts[name] = np.maximum(ins[0], 0)
        elif l["class_name"] == "Add":
            acts[name] = ins[0] + ins[1]
        elif l["class_name"] == "Flatten":
            acts[name] = ins[0].reshape(ins[0].shape[0], -1)
        elif l["class_name"] == "Dense":
            y = ins[0] @ w[name]["kernel"].astype(np.float64) + w[name]["bias"].astype(np.float64)
            if c["activation"] == "relu":
                y = np.maximum(y, 0)
            elif c["activation"] == "tanh":
                y = np.tanh(y)
            elif c["activation"] == "softmax":
                e = np.exp(y - y.max(axis=1, keepdims=True))
                y = e / e.sum(axis=1, keepdims=True)
            acts[name] = y
    return acts["policy_out"], acts["value_out"][:, 0]


def test_import_keras_weights_and_forward():
    import torch
    from cchess_alphazero.agent.model import CChessModel, InferenceNet
    from cchess_alphazero.config import Config
    cfg, w = _tiny_weights()
    m = CChessModel(Config(config_type="mini"))
    assert m.load(os.path.join(GOLD, "keras_tiny.json"), os.path.join(GOLD, "keras_tiny.h5"))
    net = m.model
    assert net.cfg["cnn_filter_num"] == 32 and net.cfg["res_layer_num"] == 2 and net.cfg["n_labels"] == 50
    assert net.cfg["policy_filters"] == 2 and net.cfg["value_filters"] == 4
    assert m.digest == m.fetch_digest(os.path.join(GOLD, "keras_tiny.h5"))
    # every tensor arrived, in torch layout
    assert np.array_equal(net.input_conv.weight.detach().numpy(), w["input_conv-5-32"]["kernel"].transpose(3, 2, 0, 1))
    assert np.array_equal(net.res[1].conv2.weight.detach().numpy(), w["res2_conv2-3-32"]["kernel"].transpose(3, 2, 0, 1))
    assert np.array_equal(net.res[0].bn1.running_var.numpy(), w["res1_batchnorm1"]["moving_variance"])
    assert np.array_equal(net.policy_out.weight.detach().numpy(), w["policy_out"]["kernel"].T)
    assert np.array_equal(net.value_out.bias.detach().numpy(), w["value_out"]["bias"])
    # and the imported network computes what the Keras graph computes
    rng = np.random.RandomState(5)
    x = (rng.rand(3, 14, 10, 9) < 0.15).astype(np.float32)
    p_ref, v_ref = _keras_forward_numpy(cfg, w, x)
    net.eval()
    with torch.no_grad():
        p, v = net(torch.from_numpy(x))
        p2, v2 = InferenceNet(net)(torch.from_numpy(x))
    assert np.abs(p.numpy() - p_ref).max() < 1e-5 and np.abs(v.numpy() - v_ref).max() < 1e-5
    assert np.abs(p2.numpy() - p_ref).max() < 1e-5 and np.abs(v2.numpy() - v_ref).max() < 1e-5


def test_import_reports_shape_mismatch(tmp_path):
    from cchess_alphazero.agent.model import CChessNet
    from cchess_alphazero.lib import keras_io
    cfg = json.load(open(os.path.join(GOLD, "keras_tiny.json")))
    kw = keras_io.config_from_keras(cfg)
    kw["cnn_filter_num"] = 64
    with pytest.raises(keras_io.KerasFormatError):
        keras_io.load_keras_weights(CChessNet(**kw), os.path.join(GOLD, "keras_tiny.h5"), keras_io.names_from_keras(cfg))
    with pytest.raises(keras_io.KerasFormatError):
        keras_io.config_from_keras({"layers": []})


def test_best_model_digest_and_reload_decision(tmp_path):
    """lib/model_helper.py (reference lib/model_helper.py:9-49): save / load the best model, reload only when the weight
    file's sha256 changed."""
    import torch
    from cchess_alphazero.agent.model import CChessModel
    from cchess_alphazero.config import Config
    from cchess_alphazero.lib import model_helper
    cfg = Config(config_type="mini")
    cfg.resource.model_best_config_path = str(tmp_path / "model_best_config.json")
    cfg.resource.model_best_weight_path = str(tmp_path / "model_best_weight.h5")
    a = CChessModel(cfg)
    a.build(seed=1)
    model_helper.save_as_best_model(a)
    b = CChessModel(cfg)
    assert model_helper.load_best_model_weight(b) and b.digest == a.digest
    assert not model_helper.need_to_reload_best_model_weight(b)
    for pa, pb in zip(a.model.parameters(), b.model.parameters()):
        assert torch.equal(pa, pb)
    c = CChessModel(cfg)
    c.build(seed=2)
    model_helper.save_as_best_model(c)                     # a newer generation lands on disk
    assert model_helper.need_to_reload_best_model_weight(b)
    assert model_helper.load_best_model_weight(b) and b.digest == c.digest
    assert not model_helper.need_to_reload_best_model_weight(b)
    # a Keras pair in the same slots is picked up through the same calls
    import shutil
    shutil.copy(os.path.join(GOLD, "keras_tiny.json"), cfg.resource.model_best_config_path)
    shutil.copy(os.path.join(GOLD, "keras_tiny.h5"), cfg.resource.model_best_weight_path)
    assert model_helper.need_to_reload_best_model_weight(b)
    assert model_helper.load_best_model_weight(b) and b.model.cfg["n_labels"] == 50
