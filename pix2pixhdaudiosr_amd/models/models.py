"""create_model(opt) with the reference's contract (models/models.py:3-20).

The reference wraps the model in nn.DataParallel when training without --fp16; here data parallelism is one
process per GPU with an RCCL all-reduce of the flat gradient buffers (parallel_state.py), so the returned object
is always the model itself; it exposes ``.module`` (returning itself) so call sites written for either the
wrapped (generate_audio.py:34) or the unwrapped (train.py:54) style work unchanged.
"""


def create_model(opt):
    if opt.model != 'pix2pixHD':
        raise NotImplementedError("only --model pix2pixHD is on the HIP hot path (UIModel is deprecated upstream)")
    from .pix2pixHD_model import Pix2PixHDModel, InferenceModel
    model = Pix2PixHDModel() if opt.isTrain else InferenceModel()
    model.initialize(opt)
    if getattr(opt, 'verbose', False):
        print("model [%s] was created" % (model.name()))
    return model
