"""``from diffusers.utils import load_image`` (/root/reference/inference_IMAGdressing_controlnetinpainting.py:17) and the two names
``adapter/attention_processor.py`` of the reference imports (:6-7)."""
USE_PEFT_BACKEND = False


def load_image(image, convert_method=None):
    """Local path or PIL image -> RGB PIL image with EXIF orientation applied (no URL fetching: there is no network client)."""
    from PIL import Image, ImageOps
    if isinstance(image, str):
        if image.startswith(("http://", "https://")):
            raise ValueError("load_image: URLs are not supported here; download the file and pass its path")
        image = Image.open(image)
    image = ImageOps.exif_transpose(image)
    return convert_method(image) if convert_method is not None else image.convert("RGB")
