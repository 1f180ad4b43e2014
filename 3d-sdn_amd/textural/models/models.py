"""create_model (reference: textural/models/models.py:5-19).  As there, a training model is wrapped in nn.DataParallel
whenever gpu ids are given -- textural/train.py:75-144 reads `model.module.*` unconditionally.  With one gpu id
DataParallel calls the module directly; with several, every replica compiles its own conv chains (networks._Fused).
The scaling path of this build is one PROCESS per GPU (sdn_hip/dist.py), not DataParallel threads."""
import torch


def create_model(opt):
    if opt.model != 'pix2pixHD':
        raise NotImplementedError('model [%s]: only pix2pixHD is part of the 3D-SDN pipeline' % opt.model)
    from .pix2pixHD_model import Pix2PixHDModel
    model = Pix2PixHDModel()
    model.initialize(opt)
    print('model [%s] was created' % model.name())
    if opt.isTrain and len(opt.gpu_ids):
        model = torch.nn.DataParallel(model, device_ids=opt.gpu_ids)
    return model
