"""Record golden vectors of the segmentation wrappers' predict() tails (build container only).

    python tests/golden/make_stage1_golden.py        -> tests/golden/stage1/*.npz

sem_objnav/segment/infer.py is executed from where it lies under /root/reference: the wrappers'
own __init__ and predict() run, so the class tables, the division by the temperature, the
softmax, the entropy, the arg-max, the schema mapping and the goal-class pooling are the
reference's code on torch (CPU) and numpy.  What is NOT there is replaced by inert doubles:
  * the modules infer.py imports at its top and the image lacks (cv2, torchvision, emsanet) and
    transformers (present, but its model classes are not wanted here) are empty import shims;
    nothing the tails compute goes through them;
  * the network is a callable that hands back the synthetic logits of the case, the
    preprocessor / feature extractor hands back placeholders, `.cuda()` / `.to()` on those
    placeholders do nothing.
This script never travels to the GPU box, nor does any reference source: only the data files
are committed.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "stage1")
INFER = "/root/reference/sem_objnav/segment/infer.py"


class Placeholder:
    """Stands for a preprocessed input tensor: the tails only move it around."""
    pixel_values = None

    def unsqueeze(self, *_):
        return self

    def cuda(self, *_):
        return self

    def to(self, *_):
        return self


Placeholder.pixel_values = Placeholder()


class FakeNet:
    """The network: returns the case's logits in the shape the wrapper expects."""
    def __init__(self):
        self.logits = None            # (C, H, W) float32 tensor of the current case
        self.style = "emsanet"

    def eval(self):
        return self

    def cuda(self, *_):
        return self

    def to(self, *_):
        return self

    def __call__(self, inputs, **kw):
        if self.style == "emsanet":
            return {"semantic_output": self.logits[None].clone()}
        return types.SimpleNamespace(logits=self.logits[None].clone())


NET = FakeNet()


def install_shims(n_emsanet_classes):
    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    module("cv2")
    tv = module("torchvision")
    tv.models = module("torchvision.models")
    tv.models.detection = module("torchvision.models.detection", MaskRCNN_ResNet50_FPN_Weights=None,
                                 maskrcnn_resnet50_fpn=None)

    class ArgParser:
        def parse_args(self, args):
            return types.SimpleNamespace(hm3d_semantic_n_classes=n_emsanet_classes)

    module("emsanet")
    module("emsanet.args", ArgParserEMSANet=ArgParser)
    module("emsanet.data", get_datahelper=lambda args: types.SimpleNamespace(dataset_config=None, datasets_valid=[None]))
    module("emsanet.model", EMSANet=lambda args, cfg: NET)
    module("emsanet.preprocessing", get_preprocessor=lambda *a, **k: (lambda obs: {"rgb": Placeholder()}))
    module("emsanet.weights", load_weights=lambda *a, **k: None)

    class Pretrained:
        @staticmethod
        def from_pretrained(name):
            return NET

    class Extractor:
        @staticmethod
        def from_pretrained(name):
            return lambda rgb, return_tensors=None: Placeholder()

    module("transformers", AutoModelForMaskGeneration=None, AutoProcessor=None,
           OneFormerForUniversalSegmentation=None, OneFormerProcessor=None,
           SegformerFeatureExtractor=Extractor, SegformerForSemanticSegmentation=Pretrained, pipeline=None)


def load_infer():
    spec = importlib.util.spec_from_file_location("ref_infer", INFER)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod          # its dataclasses look their module up
    spec.loader.exec_module(mod)
    return mod


def edge_logits(rng, c, h, w, scale, masked=-np.inf):
    """Random logits with rows of exact ties, a saturating class, masked classes and wide ranges."""
    x = (scale * rng.standard_normal((c, h, w))).astype(np.float32)
    x[:, 0, :] = 1.25
    x[:, 1, :] = 0.0
    x[4, 1, :] = 0.5
    x[7, 1, :] = 0.5
    x[:, 2, :] = -3.0
    x[5, 2, :] = 300.0
    x[1::2, 3, :] = masked
    x[:, 4, :] *= 40.0
    return x


def main():
    os.makedirs(OUT, exist_ok=True)
    install_shims(37)
    ref = load_infer()
    tmp = tempfile.mkdtemp()
    cfg_path, ckpt_path = os.path.join(tmp, "cfg.txt"), os.path.join(tmp, "ckpt.pth")
    with open(cfg_path, "w") as fh:
        fh.write("python --none\n")
    torch.save({"state_dict": {}}, ckpt_path)
    cases = [("emsanet_T1", "emsanet", 37, 1.0, 3.0), ("emsanet_T3p2", "emsanet", 37, 3.2, 3.0),
             ("emsanet_sharp", "emsanet", 37, 1.0, 8.0), ("segformer_T2p5", "segformer", 150, 2.5, 3.0),
             ("segformer_T1", "segformer", 150, 1.0, 5.0)]
    for name, style, c, temp, scale in cases:
        rng = np.random.default_rng(sum(ord(ch) for ch in name))
        h, w = 24, 32
        # SegformerWrapper resizes the network output to the image size first (infer.py:305-312): at
        # equal sizes that is the identity for finite values (checked below), but 0 * inf = NaN, so
        # its masked classes are finite
        logits = edge_logits(rng, c, h, w, scale, masked=-np.inf if style == "emsanet" else -90.0)
        NET.logits, NET.style = torch.from_numpy(logits.copy()), style
        if style == "segformer":
            same = torch.nn.functional.interpolate(NET.logits[None], size=(h, w), mode="bilinear", align_corners=False)[0]
            assert torch.equal(same, NET.logits), "the resize to the same size is not the identity"
        if style == "emsanet":
            wrapper = ref.EmsanetWrapper(ckpt_path, cfg_path, "objnav_occ_id", device="cpu", temperature=temp)
            order = list(wrapper.pred2objnav_occ_id.keys())
            table = dict(wrapper.pred2objnav_occ_id)
            out = wrapper.predict(np.zeros((h, w, 3), np.uint8), np.zeros((h, w), np.float32))
        else:
            wrapper = ref.SegformerWrapper("objnav_occ_id", device="cpu", temperature=temp)
            order = list(wrapper.pred2objnav_occ_id.keys())
            table = dict(wrapper.pred2objnav_occ_id)
            out = wrapper.predict(np.zeros((h, w, 3), np.uint8), np.zeros((h, w), np.float32))
        lut = np.asarray(wrapper.map_seg_to_schema(np.arange(c)))
        meta = dict(name=name, wrapper=style, C=c, temperature=temp, table_order=[int(k) for k in order],
                    table={str(k): int(v) for k, v in table.items()}, torch=torch.__version__, numpy=np.__version__)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), logits=logits, lut=lut.astype(np.int64),
                            semantic=np.asarray(out["semantic"]).astype(np.int32),
                            semantic_max_prob=np.asarray(out["semantic_max_prob"], dtype=np.float32),
                            semantic_prob=np.asarray(out["semantic_prob"], dtype=np.float32),
                            semantic_unc=np.asarray(out["semantic_unc"], dtype=np.float32),
                            meta=np.array(json.dumps(meta)))
        print(f"{name:18s} C={c:3d} T={temp} goal pixels={(np.asarray(out['semantic']) >= 3).sum()} "
              f"{os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
