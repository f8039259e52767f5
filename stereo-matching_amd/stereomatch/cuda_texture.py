"""Placeholder for upstream's ``stereomatch.cuda_texture`` (``CUDATexture`` wraps a cudaArray
and a texture object).  CDNA4 has no texture units -- hipcc rejects ``tex2D`` for gfx950 -- so
the texture path is out of scope; the module exists so that ``import stereomatch.cuda_texture``
keeps working and fails with a clear message only when a texture is actually requested."""


class CUDATexture:
    def __init__(self, *args, **kwargs):
        raise RuntimeError("CUDATexture needs texture hardware, which MI355X (gfx950) does not "
                           "have; pass tensors to stereomatch.cost.SSD instead")

    @classmethod
    def from_tensor(cls, *args, **kwargs):
        return cls()
