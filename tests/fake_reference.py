"""A stand-in for the reference's `models` module and for `tensorflow`, for ONE purpose: running
tests/golden/make_reference_golden.py end to end where TensorFlow does not exist, so that its plumbing -- layer
names, `set_weights` / `get_weights`, the recurrent stepping, the fixture format -- is known to work before anyone
spends a TensorFlow machine on it.  (Test infrastructure.)

It proves NOTHING about parity: the "Keras models" here compute with oracle/ju_oracle.py itself.  The layer lists
and variable shapes follow the reference constructors (scripts/training/models.py:224-252, 297-330, 378-475,
531-593); the call convention of the inference model follows `get_inference_model` (:752-829) and
keras_models.py:58-73."""

import types

import numpy as np

from joshupscale_amd import keras_import, model_file as M
from oracle import ju_oracle as O


class Layer:
    def __init__(self, name, shapes, epsilon=None):
        self.name = name
        self._w = [np.zeros(s, np.float32) for s in shapes]
        if epsilon is not None:
            self.epsilon = epsilon

    @property
    def weights(self):
        return self._w

    def get_weights(self):
        return [w.copy() for w in self._w]

    def set_weights(self, values):
        if len(values) != len(self._w):
            raise ValueError(f"{self.name}: {len(values)} arrays for {len(self._w)} variables")
        for i, v in enumerate(values):
            v = np.asarray(v, np.float32)
            if v.shape != self._w[i].shape:
                raise ValueError(f"{self.name}: shape {v.shape}, expected {self._w[i].shape}")
            self._w[i] = v.copy()


class Model:
    def __init__(self, name, layers, **meta):
        self.name = name
        # (a real Keras model also lists weight-less layers: inputs, concat, activations)
        self.layers = [Layer("input_0", [])] + layers + [Layer("a_out", [])]
        self.meta = meta

    def get_layer(self, name):
        for layer in self.layers:
            if layer.name == name:
                return layer
        raise ValueError(f"No such layer: {name}")


def _conv(name, k, cin, cout, bias=False):
    return Layer(name, [(k, k, cin, cout)] + ([(cout,)] if bias else []))


def _bn(name, c):
    return Layer(name, [(c,)] * 4, epsilon=1e-3)


def _res_blocks(n_blocks, c):
    out = []
    for i in range(1, n_blocks + 1):
        out += [_conv(f"block_{i}_conv_1", 3, c, c), _bn(f"block_{i}_bn_1", c),
                _conv(f"block_{i}_conv_2", 3, c, c), _bn(f"block_{i}_bn_2", c)]
    return out


def get_flow_resnet(num_inputs=4, num_filters=64, num_res_blocks=10, activation="relu", name="flow"):
    layers = [_conv("conv_1", 3, 3 * num_inputs, num_filters), _bn("bn_1", num_filters)]
    layers += _res_blocks(num_res_blocks, num_filters)
    layers.append(_conv("conv_2", 1, num_filters, 32, bias=True))
    return Model(name, layers, arch="resnet", activation=activation)


def get_flow_autoencoder(num_inputs=4, filters=None, activation="relu", name="flow"):
    filters = filters or [32, 64, 128, 256, 128, 64, 32]
    layers, cin = [], 3 * num_inputs
    nb = len(filters) // 2
    for i in range(1, 2 * nb + 1):
        f = filters[i - 1]
        layers += [_conv(f"block_{i}_conv_1", 3, cin, f), _bn(f"block_{i}_bn_1", f),
                   _conv(f"block_{i}_conv_2", 3, f, f), _bn(f"block_{i}_bn_2", f)]
        cin = f
    if len(filters) % 2:
        layers += [_conv("conv_1", 3, cin, filters[-1]), _bn("bn_1", filters[-1])]
        cin = filters[-1]
    layers.append(_conv("conv_2", 3, cin, 32, bias=True))
    return Model(name, layers, arch="autoencoder", activation=activation)


def get_generator_resnet(num_filters=64, num_res_blocks=24, num_fade_in_res_blocks=0, fade_in_period=0,
                         activation="relu", name="generator"):
    layers = [_conv("conv_1", 3, 51, num_filters), _bn("bn_1", num_filters)] + _res_blocks(num_res_blocks, num_filters)
    layers += [Layer("conv_trans_1", [(2, 2, 32, num_filters)]), _bn("bn_2", 32),
               Layer("conv_trans_2", [(2, 2, 3, 32), (3,)])]
    return Model(name, layers, activation=activation)


class Inference:
    def __init__(self, gen, flow, frame_height, frame_width, flow_pad_factor, normalize_brightness):
        self.gen, self.flow = gen, flow
        self.hw = (frame_height, frame_width)
        self.pad = flow_pad_factor or 0
        self.brightness = normalize_brightness
        read = lambda m: {l.name: l.get_weights() for l in m.layers if l.weights}  # noqa: E731
        gact, gslope = keras_import.activation_fields(gen.meta["activation"])
        fact, fslope = keras_import.activation_fields(flow.meta["activation"])
        base = M.ModelConfig(frame_height=frame_height, frame_width=frame_width, flow_pad_factor=self.pad,
                             normalize_brightness=normalize_brightness, flow_activation=fact, gen_activation=gact,
                             flow_negative_slope=fslope, gen_negative_slope=gslope)
        cfg, wts = keras_import.container_weights(read(gen), read(flow), base)
        self.cfg = O.ModelConfig(
            frame_height=frame_height, frame_width=frame_width, num_flow_inputs=cfg.num_flow_inputs,
            flow_arch=cfg.flow_arch, flow_filters=tuple(cfg.flow_filters), flow_res_filters=cfg.flow_res_filters,
            flow_res_blocks=cfg.flow_res_blocks, flow_pad_factor=self.pad, gen_filters=cfg.gen_filters,
            gen_blocks=cfg.gen_blocks, normalize_brightness=normalize_brightness,
            bn_eps=float(np.float32(1e-3)), flow_activation=fact, gen_activation=gact,
            flow_negative_slope=float(np.float32(fslope)), gen_negative_slope=float(np.float32(gslope)))
        self.wts = {k: np.asarray(v, np.float64) for k, v in wts.items()}
        self.inputs = [None] * (1 + cfg.num_flow_inputs)  # cur_frame, pre_gen, last_frame_0 ..

    def __call__(self, inputs, training=False):
        assert training is False
        cur, pre_gen, last = inputs[0], inputs[1], list(inputs[2:])
        assert cur.dtype == np.uint8 and cur.shape == (1,) + self.hw + (3,)
        state = O.State(np.asarray(pre_gen[0], np.float64), [np.asarray(x[0], np.float64) for x in last])
        out = O.inference_step(np.asarray(cur[0]), state, self.wts, self.cfg)
        f32 = lambda a: np.asarray(a, np.float32)[None]  # noqa: E731
        return {"output": out.output[None], "output_denorm": f32(out.output_raw), "output_raw": f32(out.state.pre_gen),
                "pre_warp": f32(out.pre_warp), "last_frames": [f32(x) for x in out.state.last_frames]}


def get_inference_model(generator_model, flow_model, skip_processing=True, frame_height=None, frame_width=None,
                        flow_pad_factor=None, normalize_brightness=False, name="inference"):
    assert skip_processing is False, "the fixture script drives the u8-in / u8-out graph"
    return Inference(generator_model, flow_model, frame_height, frame_width, flow_pad_factor, normalize_brightness)


def fake_tensorflow() -> types.ModuleType:
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "0-fake (tests/fake_reference.py)"
    tf.zeros = lambda shape, dtype=None: np.zeros(shape, np.float32)
    tf.constant = lambda x: np.asarray(x)
    return tf


# ---- `create_models` (scripts/training/models.py:1138-1194) for tools/export_jupw_from_keras.py's test ----
MODELS = {"flow-resnet": get_flow_resnet, "flow-autoencoder": get_flow_autoencoder,
          "generator-resnet": get_generator_resnet}


def create_models(config):
    """Entries {"name": <model type>, "weights": <file>, **constructor arguments}; `weights` is an .npz of
    {"<layer>/<index>": array} here (the reference loads a Keras .weights.h5 at this point)."""
    built = {}
    for key, args in config.items():
        kw = {k: v for k, v in args.items() if k not in ("name", "weights", "freeze", "copy_weights", "copy_variables")}
        if args["name"] not in MODELS:
            raise ValueError(f"Unknown model type {args['name']}")
        model = MODELS[args["name"]](name=key, **kw)
        if "weights" in args:
            data = np.load(args["weights"])
            for layer in model.layers:
                if layer.weights:
                    layer.set_weights([data[f"{layer.name}/{i}"] for i in range(len(layer.weights))])
        built[key] = model
    return built
