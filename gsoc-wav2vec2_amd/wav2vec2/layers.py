"""The Keras-inherited object surface the reference's callers touch: ``model.layers``,
``layer.trainable``, ``model.trainable``, ``layer.variables``.

``src/main.py:210`` freezes the backbone with ``model.layers[0].trainable = False``; ``:232-237`` un-freezes
everything with ``model.trainable = True`` and then freezes ``model.layers[0].layers[i]`` for the first
``len(model.layers[0].layers) - 2`` sub-layers, i.e. the 7 conv layers: Keras tracks the attributes of
``Wav2Vec2Model`` in assignment order (modeling.py:123-158) -- the ``feature_extractor`` list (flattened),
``feature_projection``, ``encoder`` -- and those of ``Wav2Vec2ForCTC`` as ``model``, ``dropout``, ``lm_head``
(modeling.py:227-229).  This module rebuilds exactly that tree over the flat variable inventory; nothing
here computes -- a layer is a named group of variables with a ``trainable`` switch.

Keras semantics kept: assigning ``layer.trainable`` also assigns every sub-layer; a variable trains only if
its own flag and the flag of every layer above it are set (``trainable_weights`` of a frozen layer is empty).
"""


class Layer:
    def __init__(self, name, prefixes=(), children=(), on_change=None, **attrs):
        self.name = name
        self._prefixes = tuple(prefixes)        # local-name prefixes of the variables this layer owns DIRECTLY
        self._children = list(children)
        self._trainable = True
        self._on_change = on_change
        self._own = []
        self.__dict__.update(attrs)

    # -- tree -------------------------------------------------------------------------------------
    @property
    def layers(self):
        return list(self._children)

    def _walk(self):
        yield self
        for c in self._children:
            yield from c._walk()

    # -- trainable --------------------------------------------------------------------------------
    @property
    def trainable(self):
        return self._trainable

    @trainable.setter
    def trainable(self, value):
        for layer in self._walk():
            layer._trainable = bool(value)
        if self._on_change is not None:
            self._on_change()

    # -- variables --------------------------------------------------------------------------------
    @property
    def variables(self):
        out = list(self._own)
        for c in self._children:
            out.extend(c.variables)
        return out

    weights = variables

    @property
    def trainable_variables(self):
        return [v for v in self.variables if v.trainable]

    @property
    def non_trainable_variables(self):
        return [v for v in self.variables if not v.trainable]

    trainable_weights = trainable_variables
    non_trainable_weights = non_trainable_variables

    def count_params(self):
        n = 0
        for v in self.variables:
            k = 1
            for d in v.shape:
                k *= int(d)
            n += k
        return n

    def __repr__(self):
        return f"<Layer {self.name} variables={len(self.variables)} trainable={self._trainable}>"


def build_backbone_layers(config, on_change):
    """Sub-layers of ``Wav2Vec2Model`` in Keras' tracking order: 7 x FeatureExtractorLayer, FeatureProjection,
    Wav2Vec2Encoder (its transformer layers as sub-layers)."""
    fe = [Layer(f"feature_extractor/conv_layers/{i}", [f"feature_extractor/conv_layers/{i}/"], on_change=on_change)
          for i in range(len(config.filter_sizes))]
    proj = Layer("feature_projection", ["feature_projection/"], on_change=on_change)
    enc_layers = [Layer(f"encoder/layers/{i}", [f"encoder/layers/{i}/"], on_change=on_change) for i in range(config.num_layers)]
    pos = Layer("encoder/pos_conv_embed", ["encoder/pos_conv_embed/"], on_change=on_change)
    enc_ln = Layer("encoder/layer_norm", ["encoder/layer_norm/"], on_change=on_change)
    enc_drop = Layer("encoder/dropout", on_change=on_change, rate=config.dropout)
    # encoder.py: pos_conv_embed, layer_norm, dropout, layers -- the attribute order of Wav2Vec2Encoder.__init__
    encoder = Layer("encoder", children=[pos, enc_ln, enc_drop] + enc_layers, on_change=on_change)
    return fe, proj, encoder


def attach(root, variables):
    """Hand every variable to the deepest layer whose prefix matches its local name and record the chain of layers
    above it (the flags that gate it)."""
    def place(layer, chain, v):
        chain = chain + [layer]
        for c in layer._children:
            if place(c, chain, v):
                return True
        if any(v.local_name.startswith(p) for p in layer._prefixes):
            layer._own.append(v)
            v._gates = chain
            return True
        return False

    for v in variables:
        if not place(root, [], v):
            raise KeyError(f"variable `{v.local_name}` belongs to no layer")
