"""Mean subtraction + bilinear resize of the input image, mirroring core/img_preprocessor.py:12-35."""
import torch

from monopsr_amd.core import device_net as dn


class ImgPreprocessor:
    _KITTI_CHANNEL_MEANS = [92.8403, 97.7996, 93.5843]
    _IMAGENET_CHANNEL_MEANS = [123.68, 116.78, 103.94]

    def preprocess_input(self, tensor_in, output_size, mean_sub_type):
        """tensor_in (batch, H, W, 3) on the GPU -> float32 (batch, output_size) mean-subtracted, resized with
        tf.image.resize_images semantics (bilinear, align_corners=False)."""
        image = tensor_in.to(torch.float32)
        if mean_sub_type == 'kitti':
            channel_means = self._KITTI_CHANNEL_MEANS
        elif mean_sub_type == 'imagenet':
            channel_means = self._IMAGENET_CHANNEL_MEANS
        else:
            raise ValueError('Invalid mean subtraction type {}'.format(mean_sub_type))
        image_centered = image - torch.tensor(channel_means, dtype=torch.float32, device=image.device)
        return dn.resize_bilinear(image_centered, tuple(output_size), align_corners=False)
