"""E2VID inference option defaults (reference: e2vid/options/inference_options.py).  Only the flags the ESS
training path consumes through ImageReconstructor / EventPreprocessor are kept."""
import argparse


def set_inference_options(parser):
    parser.add_argument('--use_gpu', dest='use_gpu', action='store_true')
    parser.set_defaults(use_gpu=True)
    parser.add_argument('--hot_pixels_file', default=None, type=str)
    parser.add_argument('--flip', dest='flip', action='store_true')
    parser.set_defaults(flip=False)
    parser.add_argument('--Imin', default=0.0, type=float)
    parser.add_argument('--Imax', default=1.0, type=float)
    parser.add_argument('--auto_hdr', dest='auto_hdr', action='store_true')
    parser.set_defaults(auto_hdr=False)
    parser.add_argument('--auto_hdr_median_filter_size', default=10, type=int)
    parser.add_argument('--unsharp_mask_amount', default=0.3, type=float)
    parser.add_argument('--unsharp_mask_sigma', default=1.0, type=float)
    parser.add_argument('--bilateral_filter_sigma', default=0.0, type=float)
    parser.add_argument('--color', dest='color', action='store_true')
    parser.set_defaults(color=False)
    parser.add_argument('--no-normalize', dest='no_normalize', action='store_true')
    parser.set_defaults(no_normalize=False)
    parser.add_argument('--no-recurrent', dest='no_recurrent', action='store_true')
    parser.set_defaults(no_recurrent=False)


def default_options():
    parser = argparse.ArgumentParser(description='E2VID.')
    set_inference_options(parser)
    return parser.parse_known_args([])[0]
