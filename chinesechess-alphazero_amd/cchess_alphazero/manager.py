"""Command line of ``run.py`` (reference: cchess_alphazero/manager.py).  ``self`` (the hot path) and ``eval``
(the arena) are served by this package; the other sub-commands of the reference (opt, play, sl, ob)
are outside the scope table and report that."""
import argparse
import os
from logging import getLogger

from cchess_alphazero.config import Config, PlayWithHumanConfig
from cchess_alphazero.lib.logger import setup_logger

logger = getLogger(__name__)

CMD_LIST = ['self', 'opt', 'eval', 'play', 'eval', 'sl', 'ob']
PIECE_STYLE_LIST = ['WOOD', 'POLISH', 'DELICATE']
BG_STYLE_LIST = ['CANVAS', 'DROPS', 'GREEN', 'QIANHONG', 'SHEET', 'SKELETON', 'WHITE', 'WOOD']
RANDOM_LIST = ['none', 'small', 'medium', 'large']


def create_parser():
    p = argparse.ArgumentParser()
    p.add_argument("cmd", help="what to do", choices=CMD_LIST)
    p.add_argument("--new", help="run from new best model", action="store_true")
    p.add_argument("--type", help="use normal setting", default="mini")
    p.add_argument("--total-step", help="set TrainerConfig.start_total_steps", type=int)
    p.add_argument("--ai-move-first", help="set human or AI move first", action="store_true")
    p.add_argument("--cli", help="play with AI with CLI, default with GUI", action="store_true")
    p.add_argument("--gpu", help="device list", default="0")
    p.add_argument("--onegreen", help="train sl work with onegreen data", action="store_true")
    p.add_argument("--skip", help="skip games", default=0, type=int)
    p.add_argument("--ucci", help="play with ucci engine instead of self play", action="store_true")
    p.add_argument("--piece-style", help="choose a style of piece", choices=PIECE_STYLE_LIST, default="WOOD")
    p.add_argument("--bg-style", help="choose a style of board", choices=BG_STYLE_LIST, default="WOOD")
    p.add_argument("--random", help="choose a style of randomness", choices=RANDOM_LIST, default="none")
    p.add_argument("--distributed", help="whether upload/download file from remote server", action="store_true")
    p.add_argument("--elo", help="whether to compute elo score", action="store_true")
    # engine knobs (not in the reference)
    p.add_argument("--games-per-gpu", type=int, default=None, help="concurrent games per GPU")
    p.add_argument("--net-dtype", default=None, choices=["float32", "bfloat16", "float16"])
    p.add_argument("--max-rounds", type=int, default=None, help="stop after this many lock-step rounds (default: never)")
    p.add_argument("--max-games", type=int, default=None, help="stop after this many finished games (default: never)")
    return p


def setup(config, args):
    config.opts.new = args.new
    if args.total_step is not None:
        config.trainer.start_total_steps = args.total_step
    config.opts.device_list = args.gpu
    config.resource.create_directories()
    if args.cmd == 'self':
        setup_logger(config.resource.play_log_path)
    else:
        setup_logger(config.resource.main_log_path)


def start():
    args = create_parser().parse_args()
    config = Config(config_type=args.type)
    if args.games_per_gpu:
        config.engine.games_per_gpu = args.games_per_gpu
    if args.net_dtype:
        config.engine.net_dtype = args.net_dtype
    config.engine.max_rounds = args.max_rounds
    config.engine.max_games = args.max_games
    config.opts.piece_style = args.piece_style
    config.opts.bg_style = args.bg_style
    config.internet.distributed = args.distributed
    if len(args.gpu.split(',')) > 1:                      # reference manager.py:66-70
        config.opts.use_multiple_gpus = True
        config.opts.gpu_num = len(args.gpu.split(','))
        logger.info(f"User GPU {args.gpu}")
    setup(config, args)
    logger.info('Config type: %s' % (args.type))
    if args.cmd == 'self':
        if args.ucci:
            raise SystemExit("self-play against an external UCCI engine is outside the MI355X hot path")
        from cchess_alphazero.worker import self_play
        return self_play.start(config)
    if args.cmd == 'eval':                                   # reference manager.py:94-103
        if args.elo:
            raise SystemExit("the server-driven Elo evaluator needs cczero.org (no network): outside the hot path")
        config.eval.update_play_config(config.play)
        config.opts.evaluate = True
        from cchess_alphazero.worker import evaluator
        return evaluator.start(config)
    raise SystemExit(f"`run.py {args.cmd}` is not part of the MI355X self-play hot path (SURVEY 8): "
                     f"use the reference implementation for it; the play records written by `run.py self` "
                     f"are in the reference's format")
