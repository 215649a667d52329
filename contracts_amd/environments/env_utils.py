"""Frame export for rendered episodes — the reference's `make_video_from_rgb_imgs` (environments/env_utils.py:28-58,
called from run_render.py:66-76).  The reference writes an mp4 through cv2; cv2 is used here too when it is importable,
otherwise the same frames (nearest-neighbour upscaled, as cv2.INTER_NEAREST) go out as an animated GIF through PIL."""
import os

import numpy as np


def _resize_nearest(img, size):
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_NEAREST) for integer and fractional factors"""
    width, height = size
    h, w = img.shape[:2]
    rows = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)  # floor(dst * scale): cv2's nearest rule
    cols = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return img[rows][:, cols]


def make_video_from_rgb_imgs(rgb_arrs, vid_path, video_name="trajectory", fps=5, format="mp4v", resize=None, verbose=False):
    """Create a video from a list of rgb arrays (H x W x 3).  Returns the path written."""
    if verbose:
        print("Rendering video...")
    os.makedirs(vid_path, exist_ok=True)
    if resize is not None:
        width, height = resize
    else:
        height, width, _ = rgb_arrs[0].shape
        resize = width, height
    frames = [_resize_nearest(np.asarray(im).astype(np.uint8), resize) for im in rgb_arrs]
    try:
        import cv2  # the reference's writer
        path = os.path.join(vid_path, video_name + ".mp4")
        video = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*format), float(fps), (width, height))
        for im in frames:
            video.write(im)
        video.release()
        return path
    except ImportError:
        from PIL import Image
        path = os.path.join(vid_path, video_name + ".gif")
        imgs = [Image.fromarray(im, "RGB") for im in frames]
        imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=int(round(1000.0 / fps)), loop=0, optimize=False)
        return path
