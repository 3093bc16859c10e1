"""Command-line options -- same flags, defaults and derived fields as the reference's
code/options.py:4-52 (kept as the entry-point contract), plus MI355X-side additions that default to
the reference behaviour: --patch (crop edge, reference hard-codes 80), --dtype, --synthetic."""
import argparse
import os


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--model', default='mmformer', type=str, help='model name')
    p.add_argument('-batch_size', '--batch_size', default=1, type=int, help='Batch size')
    p.add_argument('--lr', default=2e-4, type=float, help='base learning rate')
    p.add_argument('--weight_decay', default=1e-4, type=float)
    p.add_argument('--num_epochs', default=300, type=int, help='training epochs')
    p.add_argument('--temp', default=4.0, type=float, help='knowledge-distillation temperature')
    p.add_argument('--region_fusion_start_epoch', default=0, type=int, help='warm-up epochs used in rfnet')
    p.add_argument('--seed', default=1037, type=int, help='random seed')
    p.add_argument('--gpu', type=str, default='3', help='GPU to use (ignored under torchrun: LOCAL_RANK wins)')
    p.add_argument('--mask_type', default='idt', type=str, help='training settings: pdt idt or idt_drop')
    p.add_argument('--use_pretrain', action='store_true', default=False, help='whether use pretrained model')
    p.add_argument('--use_passion', action='store_true', default=False, help='whether use passion')
    p.add_argument('--use_valid', action='store_true', default=False, help='whether use validation')
    p.add_argument('--dataname', default='BraTS/BRATS2020', type=str, help='stored dataset name')
    p.add_argument('--datapath', default='BraTS/BRATS2020_Training_none_npy', type=str,
                   help='stored dataset path (ralative to datarootPath)')
    p.add_argument('--imbmrpath', default='BraTS/brats_split/Brats2020_imb_split_mr2468.csv', type=str, help='csv path')
    p.add_argument('--savepath', default='outputs/idt_mr2468_mmformer_passion_bs1_epoch300_lr2e-4_temp4', type=str,
                   help='output path')
    p.add_argument('--resume', default=None, type=str, help='pretrained model path')
    # ---- additions (defaults = reference behaviour)
    p.add_argument('--patch', default=80, type=int, help='training crop edge (multiple of 16)')
    p.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'], help='activation storage dtype')
    p.add_argument('--synthetic', action='store_true', default=False, help='synthetic volumes instead of the npy dataset')
    p.add_argument('--iters_per_epoch', default=0, type=int, help='synthetic mode: iterations per epoch')
    p.add_argument('--stop_after_epoch', default=-1, type=int,
                   help='leave after the checkpoint of this epoch (0-based) is written; --resume continues the run')
    return p


def args_parser(argv=None):
    args = build_parser().parse_args(argv)
    here = os.path.dirname(os.path.abspath(__file__))
    args.datarootPath = os.path.abspath(os.path.join(here, '../datasets'))
    args.datasetPath = os.path.abspath(os.path.join(args.datarootPath, args.datapath))
    edge = args.patch
    args.train_transforms = (f'Compose([RandCrop3D(({edge},{edge},{edge})), RandomRotion(10), '
                             'RandomIntensityChange((0.1,0.1)), RandomFlip(0), NumpyType((np.float32, np.int64)),])')
    args.test_transforms = 'Compose([NumpyType((np.float32, np.int64)),])'
    return args
