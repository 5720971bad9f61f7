from .msdeformattn import MSDeformAttnPixelDecoder, build_pixel_decoder  # noqa: F401
