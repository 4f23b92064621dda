"""Mirror of segment_anything/segment_anything/dinov2_utils.py for the hot path:
``load_dinov2_model``, ``set_torch_image``, ``get_cls_token_torch`` (same names, arguments and
error behaviour), backed by the HIP ViT in pope_amd.dinov2.
"""
import os

import numpy as np
import torch
from PIL import Image

from .dinov2 import build_vits14
from .synth import IMAGENET_MEAN, IMAGENET_STD

DEFAULT_WEIGHTS = "weights/dinov2_vits14.pth"  # dinov2_utils.py:45


def load_dinov2_weights(model, pretrained_weights, checkpoint_key="student"):
    """dinov2/dinov2/utils/utils.py:21-34: take `checkpoint_key` if present, strip `module.` /
    `backbone.` prefixes, load with strict=True."""
    state_dict = torch.load(pretrained_weights, map_location="cpu")
    if checkpoint_key is not None and checkpoint_key in state_dict:
        state_dict = state_dict[checkpoint_key]
    state_dict = {k.replace("module.", ""): v for k, v in state_dict.items()}
    state_dict = {k.replace("backbone.", ""): v for k, v in state_dict.items()}
    return model.load_state_dict(state_dict, strict=True)


def load_dinov2_model(weights=DEFAULT_WEIGHTS, state_dict=None):
    """dinov2_utils.py:38-47: ViT-S/14 (img_size 518, LayerScale 1e-5, mlp FFN), weights loaded
    strictly, eval mode, returned on CPU (the caller moves it to 'cuda:0',
    eval_linemod_json.py:11-12).  `state_dict` lets callers without the checkpoint file (offline
    benchmarks) supply weights in the same layout."""
    model = build_vits14()
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    else:
        if not os.path.exists(weights):
            raise FileNotFoundError(weights)
        load_dinov2_weights(model, weights, checkpoint_key="student")
    model.eval()
    return model


def _prep(image, resize, crop):
    # transforms.ToPILImage -> Resize -> [CenterCrop] -> ToTensor -> Normalize   (dinov2_utils.py:61-75)
    if isinstance(image, torch.Tensor):
        image = image.numpy()
    pil = Image.fromarray(np.ascontiguousarray(image))
    pil = pil.resize((resize[1], resize[0]), Image.BILINEAR)
    if crop is not None:
        w, h = pil.size
        top, left = int(round((h - crop[0]) / 2.0)), int(round((w - crop[1]) / 2.0))
        pil = pil.crop((left, top, left + crop[1], top + crop[0]))
    arr = np.asarray(pil, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (t - mean) / std


def set_torch_image(image: np.ndarray, image_format: str = "RGB", center_crop=False):
    """dinov2_utils.py:55-78.  HWC uint8 (the drivers pass cv2 BGR as-is) -> [1,3,H,W] on cuda.
    The uint8 frame is uploaded as it is and resized / cropped / normalised on the GPU (pope_amd/preprocess.py),
    bit-identical to the PIL + torchvision host path (`_prep` is that host path, kept as the parity reference);
    `set_torch_images` there takes all proposals of a query at once."""
    from .preprocess import set_torch_images
    if isinstance(image, torch.Tensor):
        image = image.numpy()
    image = np.asarray(image)
    if image.ndim == 2:
        image = np.stack([image] * 3, -1)
    return set_torch_images(image[None], center_crop=center_crop)


def get_cls_token_torch(model, input_tensor):
    """dinov2_utils.py:106-111."""
    out = model(input_tensor, is_training=True)
    return out["x_norm_clstoken"]
