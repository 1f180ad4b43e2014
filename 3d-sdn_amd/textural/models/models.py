"""create_model (reference: textural/models/models.py).  The reference wraps a training model in nn.DataParallel (one
Python thread per GPU); on MI355X the scaling path is one process per GPU (torch.distributed over RCCL, see
sdn_hip/dist.py), so the wrapper is only applied when more than one gpu id is requested."""
import torch


def create_model(opt):
    if opt.model != 'pix2pixHD':
        raise NotImplementedError('model [%s]: only pix2pixHD is part of the 3D-SDN pipeline' % opt.model)
    from .pix2pixHD_model import Pix2PixHDModel
    model = Pix2PixHDModel()
    model.initialize(opt)
    print('model [%s] was created' % model.name())
    if opt.isTrain and len(opt.gpu_ids) > 1:
        model = torch.nn.DataParallel(model, device_ids=opt.gpu_ids)
    return model
