"""Golden vectors of the reference's inference-side tensor code, written by running the REFERENCE itself (imported from /root/reference in
the build container through tools/ref_harness.py; cv2 - imported by engine_loc.py and cams_deit.py at module level, never called on this
path - is stubbed as an empty module):

  engine_loc.decouple_output (engine_loc.py:99-124): the flip test-time-augmentation merge, on seeded tensors, batch sizes 1 and 2,
  with and without aux_outputs.

-> tests/golden/infer.pt (data only: the input dicts and the reference's output dicts)."""
import copy
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.ref_harness import install_shims  # noqa: E402


def main():
    install_shims()
    for name in ("cv2", "pycocotools", "pycocotools.coco", "pycocotools.cocoeval", "pycocotools.mask"):
        sys.modules.setdefault(name, types.ModuleType(name))
    timm = sys.modules["timm"]
    if not hasattr(timm, "data"):
        timm.data = types.SimpleNamespace(IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    import engine_loc as ref
    g = torch.Generator().manual_seed(2025)
    cases = []
    for bs, Q, Kc, naux in ((2, 7, 5, 1), (1, 10, 21, 0), (3, 40, 21, 2)):
        mk = lambda *s: torch.randn(*s, generator=g)
        outp = {"pred_logits": mk(2 * bs, Q, Kc), "pred_boxes": torch.rand(2 * bs, Q, 4, generator=g), "x_logits": mk(2 * bs, Kc),
                "x_cls_logits": mk(2 * bs, Kc), "cams_cls": mk(2 * bs, Kc, 3, 4), "x_patch_unrelated": mk(2 * bs, 3)}
        if naux:
            outp["aux_outputs"] = [{"pred_logits": mk(2 * bs, Q, Kc), "pred_boxes": torch.rand(2 * bs, Q, 4, generator=g)} for _ in range(naux)]
        inp = copy.deepcopy(outp)
        res = ref.decouple_output(copy.deepcopy(outp), bs=bs)
        cases.append({"bs": bs, "input": inp, "output": res})
    path = os.path.join(ROOT, "tests", "golden", "infer.pt")
    torch.save({"decouple_output": cases}, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
