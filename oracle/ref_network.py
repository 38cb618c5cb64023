"""Import the REFERENCE network (models/networks_united.py) on CPU, in this container only.

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box: this module is
used by tests/golden/make_golden.py (fixture generation) and by CPU tests that are skipped
when the reference is absent.  Shims:
  1. ``torchvision.models.utils.load_state_dict_from_url`` does not exist here and
     models/networks_img.py:17 asks for pretrained=True (a download): a fake module is
     registered and models.resnet.resnet34 is wrapped to force pretrained=False.
  2. ``import index_max`` (models/networks_pc.py:12) resolves to a module whose
     forward_cuda_shared_mem forwards to the reference's OWN forward_cpu (oracle/_ref).
"""
import os
import sys
import types

REF = os.environ.get("DEEPI2P_REFERENCE", "/root/reference")


def available():
    from .ref_loader import load_ref_index_max
    return os.path.isdir(os.path.join(REF, "models")) and load_ref_index_max() is not None


def make_opt(N, H, W, is_fine, B=1, Ma=128, Mb=128, k_ab=16):
    import torch
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from kitti.options import Options
    opt = Options()                      # kitti/options.py:49 only constructs a device object
    opt.device = torch.device("cpu")
    opt.input_pt_num = N
    opt.img_H, opt.img_W = H, W
    opt.is_fine_resolution = is_fine
    opt.node_a_num, opt.node_b_num = Ma, Mb
    opt.k_ab = k_ab
    opt.batch_size = B
    return opt


def load_reference_detector(opt):
    import torch
    from .ref_loader import load_ref_index_max
    if REF not in sys.path:
        sys.path.insert(0, REF)
    ref_im = load_ref_index_max()
    shim = types.ModuleType("index_max")
    shim.forward_cpu = ref_im.forward_cpu
    shim.forward_cuda_shared_mem = lambda data, index, K: ref_im.forward_cpu(
        data.contiguous(), index.contiguous(), K)
    shim.forward_cuda = shim.forward_cuda_shared_mem
    sys.modules["index_max"] = shim
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvu = types.ModuleType("torchvision.models.utils")
        tvu.load_state_dict_from_url = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no network"))
        tv.models = tvm
        tvm.utils = tvu
        sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvu})
    from models import resnet as ref_resnet
    if not getattr(ref_resnet, "_oracle_wrapped", False):
        orig = ref_resnet.resnet34
        ref_resnet.resnet34 = lambda in_channels=3, pretrained=False, progress=True, **kw: orig(
            in_channels=in_channels, pretrained=False, progress=progress, **kw)
        ref_resnet._oracle_wrapped = True
    from models.networks_united import KeypointDetector
    # torch.cuda.device(-1) (networks_pc.py:88) must be a no-op on a CPU-only build
    import contextlib
    torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
    det = KeypointDetector(opt)
    det.eval()
    return det
