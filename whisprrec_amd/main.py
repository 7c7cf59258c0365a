"""Stand-alone launcher with the reference's command line (reference src/main.py) for the models of this package:

    python -m whisprrec_amd.main --model_name BPRMF --dataset ml-100k --path <data dir>/ --lr 1e-3 --l2 1e-6

Same flow as the reference: pick model / reader / runner classes by name (main.py:104-114), chain their ``parse_*_args``
(:117-122), derive the log / checkpoint names (:125-133), seed (utils.init_seed), read the corpus, build the model, train
with evaluation on dev every epoch, load the best checkpoint, report test metrics (:30-86).  The corpus pickle cache of
the reference (:53-63) is not reproduced: the NumPy reader takes a fraction of a second.  Needs a GPU (there is no CPU path).
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

from . import bprmf, lightgcn, reader, runner, sasrec, sgl

MODELS = {"BPRMF": bprmf.BPRMF, "LightGCN": lightgcn.LightGCN, "SGL": sgl.SGL, "SASRec": sasrec.SASRec}
READERS = {"BaseReader": reader.BaseReader, "SeqReader": reader.SeqReader}
RUNNERS = {"BaseRunner": runner.BaseRunner, "HipRunner": runner.HipRunner}


def parse_global_args(parser):
    parser.add_argument("--gpu", type=str, default="0", help="Set CUDA_VISIBLE_DEVICES")
    parser.add_argument("--verbose", type=int, default=logging.INFO, help="Logging Level, 0, 10, ..., 50")
    parser.add_argument("--log_file", type=str, default="", help="Logging file path")
    parser.add_argument("--random_seed", type=int, default=3407, help="Random seed of numpy and pytorch")
    parser.add_argument("--load", type=int, default=0, help="Whether load model and continue to train")
    parser.add_argument("--train", type=int, default=1, help="To train the model or not.")
    parser.add_argument("--regenerate", type=int, default=1, help="Accepted for compatibility (no corpus cache here)")
    return parser


def init_seed(seed):
    """reference utils.init_seed (src/utils/utils.py:13-20)"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_args(argv=None):
    init_parser = argparse.ArgumentParser(description="Model")
    init_parser.add_argument("--model_name", type=str, default="BPRMF", help="Choose a model to run: " + ", ".join(MODELS))
    init_parser.add_argument("--reader_name", type=str, default=None, help="Choose a reader object.")
    init_parser.add_argument("--runner_name", type=str, default=None, help="Choose a runner object: " + ", ".join(RUNNERS))
    init_args, _ = init_parser.parse_known_args(argv)
    if init_args.model_name not in MODELS:
        raise SystemExit("unknown --model_name %r (available: %s)" % (init_args.model_name, ", ".join(MODELS)))
    model_class = MODELS[init_args.model_name]
    reader_class = READERS[model_class.reader if init_args.reader_name is None else init_args.reader_name]
    runner_class = RUNNERS[model_class.runner if init_args.runner_name is None else init_args.runner_name]
    parser = argparse.ArgumentParser(description="")
    parser = parse_global_args(parser)
    parser = reader_class.parse_reader_args(parser)
    parser = runner_class.parse_runner_args(parser)
    parser = model_class.parse_model_args(parser)
    args, _ = parser.parse_known_args(argv)
    log_args = [init_args.model_name, args.dataset, str(args.random_seed)]
    for arg in ["lr", "l2"] + model_class.extra_log_args:
        log_args.append(arg + "=" + str(getattr(args, arg)))
    name = "__".join(log_args).replace(" ", "__")
    if args.log_file == "":
        args.log_file = "../log/{}/{}.txt".format(init_args.model_name, name)
    if args.model_path == "":
        args.model_path = "../model/{}/{}.pt".format(init_args.model_name, name)
    return args, model_class, reader_class, runner_class


def main(argv=None):
    args, model_class, reader_class, runner_class = build_args(argv)
    os.makedirs(os.path.dirname(os.path.abspath(args.log_file)), exist_ok=True)
    logging.basicConfig(filename=args.log_file, level=args.verbose, force=True)
    logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
    init_seed(args.random_seed)
    os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu
    if not torch.cuda.is_available():
        raise SystemExit("whisprrec_amd needs a GPU: the hot path has no CPU fallback")
    args.device = torch.device("cuda")
    logging.info("Device: {}".format(args.device))
    corpus = reader_class(args).corpus()
    model = model_class(args, corpus).to(args.device)
    logging.info("#params: {}".format(model.count_variables()))
    logging.info(model)
    data_dict = {phase: model_class.Dataset(model, corpus, phase) for phase in ("train", "dev", "test")}
    run = runner_class(args)
    if args.load > 0:
        model.load_model()
    if args.train > 0:
        run.train(data_dict)
    res = run.print_res(data_dict["test"])
    logging.info(os.linesep + "Test After Training: " + res)
    model.actions_after_train()
    return res


if __name__ == "__main__":
    main()
