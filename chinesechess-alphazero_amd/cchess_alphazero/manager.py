"""Command line of ``run.py`` -- same sub-commands and flags as the reference's cchess_alphazero/manager.py.
``self`` (the hot path) and ``eval`` (the arena) are served by the MI355X engine; the other
sub-commands of the reference (opt, play, sl, ob) are outside the scope table and say so."""
import argparse
from logging import getLogger

from cchess_alphazero.config import Config
from cchess_alphazero.lib.logger import setup_logger

logger = getLogger(__name__)

COMMANDS = ('self', 'opt', 'eval', 'play', 'sl', 'ob')
PIECE_STYLES = ('WOOD', 'POLISH', 'DELICATE')
BOARD_STYLES = ('CANVAS', 'DROPS', 'GREEN', 'QIANHONG', 'SHEET', 'SKELETON', 'WHITE', 'WOOD')
RANDOMNESS = ('none', 'small', 'medium', 'large')

# (flag, kwargs) -- the reference's options first (manager.py:16-33), then the engine's own
_FLAGS = [
    ("--new", dict(action="store_true", help="run from new best model")),
    ("--type", dict(default="mini", help="configuration: mini / normal / distribute")),
    ("--total-step", dict(type=int, help="TrainerConfig.start_total_steps")),
    ("--ai-move-first", dict(action="store_true", help="(play) the AI moves first")),
    ("--cli", dict(action="store_true", help="(play) command-line board")),
    ("--gpu", dict(default="0", help="comma separated device list; one engine process per device")),
    ("--onegreen", dict(action="store_true", help="(sl) onegreen data")),
    ("--skip", dict(default=0, type=int, help="(sl) skip games")),
    ("--ucci", dict(action="store_true", help="(self) play against a UCCI engine")),
    ("--piece-style", dict(choices=PIECE_STYLES, default="WOOD")),
    ("--bg-style", dict(choices=BOARD_STYLES, default="WOOD")),
    ("--random", dict(choices=RANDOMNESS, default="none")),
    ("--distributed", dict(action="store_true", help="upload / download through the cczero server")),
    ("--elo", dict(action="store_true", help="(eval) server-driven Elo evaluation")),
    ("--games-per-gpu", dict(type=int, help="engine: concurrent games per GPU")),
    ("--net-dtype", dict(choices=["float32", "bfloat16", "float16"], help="engine: network precision")),
    ("--max-rounds", dict(type=int, help="engine: stop after this many lock-step rounds (default: never)")),
    ("--max-games", dict(type=int, help="engine: stop after this many finished games (default: never)")),
]


def create_parser():
    parser = argparse.ArgumentParser(description="Xiangqi AlphaZero self-play on MI355X")
    parser.add_argument("cmd", choices=COMMANDS, help="what to do")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def build_config(args):
    config = Config(config_type=args.type)
    opts, engine = config.opts, config.engine
    opts.new = args.new
    opts.piece_style, opts.bg_style = args.piece_style, args.bg_style
    opts.device_list = args.gpu
    n_dev = len(args.gpu.split(','))
    if n_dev > 1:                                           # reference manager.py:66-70
        opts.use_multiple_gpus, opts.gpu_num = True, n_dev
    config.internet.distributed = args.distributed
    if args.total_step is not None:
        config.trainer.start_total_steps = args.total_step
    if args.games_per_gpu:
        engine.games_per_gpu = args.games_per_gpu
    if args.net_dtype:
        engine.net_dtype = args.net_dtype
    engine.max_rounds, engine.max_games = args.max_rounds, args.max_games
    return config


def start():
    args = create_parser().parse_args()
    config = build_config(args)
    config.resource.create_directories()
    rc = config.resource
    setup_logger(rc.play_log_path if args.cmd == 'self' else (rc.eval_log_path if args.cmd == 'eval' else rc.main_log_path))
    logger.info('Config type: %s' % (args.type))
    if args.cmd == 'self':
        if args.ucci:
            raise SystemExit("self-play against an external UCCI engine is outside the MI355X hot path")
        from cchess_alphazero.worker import self_play
        return self_play.start(config)
    if args.cmd == 'eval':                                  # reference manager.py:94-103
        if args.elo:
            raise SystemExit("the server-driven Elo evaluator needs cczero.org (no network): outside the hot path")
        config.eval.update_play_config(config.play)
        # (config.opts.evaluate stays False here, as in the reference: only its Elo evaluator sets it,
        #  compute_elo.py:88 -- so a repeated position is still played at tau = 0.5, player.py:460-461)
        from cchess_alphazero.worker import evaluator
        return evaluator.start(config)
    raise SystemExit(f"`run.py {args.cmd}` is not part of the MI355X self-play hot path (SURVEY 8): use the reference "
                     f"implementation for it; the play records written by `run.py self` are in the reference's format")
