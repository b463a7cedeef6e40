"""Configuration objects with the reference's attribute names (cchess_alphazero/config.py and
configs/{mini,normal,distribute}.py), so ``Config(type).play.simulation_num_per_move`` etc. keep working.

The per-type values live in one table (``configs/_tables.py``); the engine-specific knobs (concurrent
games per GPU, network dtype, arena size) are extra attributes of ``config.engine`` with defaults.
"""
import getpass
import os


def _project_dir():
    d = os.path.dirname
    return d(d(os.path.abspath(__file__)))


class _Section:
    """Plain attribute bag; ``update_play_config`` copies the evaluation overrides like the reference."""

    _copy_to_play = ()

    def __init__(self, **values):
        self.__dict__.update(values)

    def update_play_config(self, pc):
        for k in self._copy_to_play:
            if hasattr(self, k):
                setattr(pc, k, getattr(self, k))

    def __repr__(self):
        return f"{type(self).__name__}({self.__dict__})"


class PlayConfig(_Section):
    pass


class PlayDataConfig(_Section):
    pass


class TrainerConfig(_Section):
    pass


class ModelConfig(_Section):
    pass


class EvaluateConfig(_Section):
    _copy_to_play = ("simulation_num_per_move", "thinking_loop", "c_puct", "tau_decay_rate", "noise_eps",
                     "max_game_length", "max_processes", "search_threads")


class PlayWithHumanConfig(_Section):
    _copy_to_play = ("simulation_num_per_move", "c_puct", "noise_eps", "tau_decay_rate", "search_threads",
                     "dirichlet_alpha")

    def __init__(self):
        super().__init__(simulation_num_per_move=800, c_puct=1, search_threads=10, noise_eps=0,
                         tau_decay_rate=0, dirichlet_alpha=0.2)


class EngineConfig(_Section):
    """MI355X engine knobs (not in the reference)."""

    def __init__(self):
        super().__init__(games_per_gpu=4096,      # concurrent games = wavefronts per GPU
                         sims_per_round=None,     # lock-step batch per game; None = play.search_threads
                         net_dtype="float32",     # float32 (reference precision) | bfloat16 | float16
                         net_trunk="mfma",        # mfma (hand-written convolution kernel) | library (MIOpen)
                         net_arith="c6",          # REQUESTED products of the float32 tower: c6 (fp16 + two scaled-bf6
                                                  # correction MFMAs; 128 / 192 filters, >= 2 blocks; elsewhere it means c8) |
                                                  # c8 (e4m3 corrections; 128 / 192 filters) | c8>N (first N blocks) |
                                                  # f16x3 | bf16x3 (three MFMAs on fp16 / bf16 pairs); CZ_TOWER_ARITH overrides
                         arith_guard=True,        # measure the request against float64 on calibration positions when
                                                  # weights are loaded and fall back c6 -> c6>N -> c8 -> c8>N -> f16x3 ->
                                                  # bf16x3 -> fp32 library trunk beyond 5e-5 on policy / value / legal priors
                                                  # or 2e-4 on the logits (agent/model.py guarded_inference_net); False skips
                                                  # the candidate comparison only -- c6 still measures its images' exponents
                         audit_every_rounds=50000,  # self-play re-measures the running arithmetic on LIVE queue positions
                         audit_first_round=400,     # this often (and once early); outside the guard -> next more exact
                                                  # arithmetic (engine.audit_network; None: off)
                         max_nodes_per_game=0,    # sizes a game's hash / chunk table; 0 = the longest game's whole tree
                         pool_chunks=0,           # tree memory for all games in MiB; 0 = auto (<= 80 % of free HBM)
                         pool_fraction=None,      # with pool_chunks = 0: that fraction of the free HBM instead of 80 %
                                                  # (several processes on one GPU: max_processes > 1, a trainer, UCI)
                         max_depth=0,
                         reload_seconds=600,      # self-play re-checks the best-model digest this often (api.py:37-44)
                         compact_queue=True,      # evaluate only the queue slots that hold a new leaf (cz_search_round_q)
                         policy_logits=True,      # the engine's queue carries raw logits: no softmax pass over all 2086
                                                  # columns, the priors come from the legal moves' logits (cz_search_policy_logits)
                         leaf_masks=True,         # the search kernel also writes every leaf as a 96-word occupancy board, which the
                                                  # first block's fused input layer takes instead of scanning the planes
                         use_hip_graph=False, base_seed=0, report_every_rounds=200,
                         max_rounds=None, max_games=None)   # None = run forever, like the reference


class Options:
    new = False
    light = True
    device_list = '0'
    bg_style = 'CANVAS'
    piece_style = 'WOOD'
    random = 'none'
    log_move = False
    use_multiple_gpus = False
    gpu_num = 1
    evaluate = False
    has_history = False


class ResourceConfig:
    def __init__(self):
        env, j = os.environ.get, os.path.join
        self.project_dir = env("PROJECT_DIR", _project_dir())
        self.data_dir = env("DATA_DIR", j(self.project_dir, "data"))
        self.model_dir = env("MODEL_DIR", j(self.data_dir, "model"))
        for stem in ("model_best", "sl_best"):
            setattr(self, f"{stem}_config_path", j(self.model_dir, f"{stem}_config.json"))
            setattr(self, f"{stem}_weight_path", j(self.model_dir, f"{stem}_weight.h5"))
        self.eleeye_path = j(self.model_dir, 'ELEEYE')
        self.next_generation_model_dir = j(self.model_dir, "next_generation")
        self.next_generation_config_path = j(self.next_generation_model_dir, "next_generation_config.json")
        self.next_generation_weight_path = j(self.next_generation_model_dir, "next_generation_weight.h5")
        self.rival_model_config_path = j(self.model_dir, "rival_config.json")
        self.rival_model_weight_path = j(self.model_dir, "rival_weight.h5")
        self.play_data_dir = j(self.data_dir, "play_data")
        self.play_data_filename_tmpl = "play_%s.json"
        self.self_play_game_idx_file = j(self.data_dir, "play_data_idx")
        self.play_record_filename_tmpl = "record_%s.qp"
        self.play_record_dir = j(self.data_dir, "play_record")
        self.log_dir = j(self.project_dir, "logs")
        for name in ("main", "opt", "play", "sl", "eval"):
            setattr(self, f"{name}_log_path", j(self.log_dir, f"{name}.log"))
        self.sl_data_dir = j(self.data_dir, "sl_data")
        self.sl_data_gameinfo = j(self.sl_data_dir, "gameinfo.csv")
        self.sl_data_move = j(self.sl_data_dir, "moves.csv")
        self.sl_onegreen = j(self.sl_data_dir, "onegreen.json")
        self.font_path = j(self.project_dir, 'cchess_alphazero', 'play_games', 'PingFang.ttc')

    def create_directories(self):
        for d in (self.project_dir, self.data_dir, self.model_dir, self.play_data_dir, self.log_dir,
                  self.play_record_dir, self.next_generation_model_dir, self.sl_data_dir):
            os.makedirs(d, exist_ok=True)


class InternetConfig:
    def __init__(self):
        self.distributed = False
        self.username = getpass.getuser()
        self.base_url = 'https://cczero.org'
        self.upload_url = f'{self.base_url}/api/upload_game_file/192x10'
        self.upload_eval_url = f'{self.base_url}/api/upload_eval_game_file'
        self.download_url = 'http://download.52coding.com.cn/192x10/model_best_weight.h5'
        self.get_latest_digest = f'{self.base_url}/api/get_latest_digest/192x10'
        self.add_model_url = f'{self.base_url}/api/add_model'
        self.get_evaluate_model_url = f'{self.base_url}/api/query_for_evaluate'
        self.download_base_url = 'http://download.52coding.com.cn/'
        self.get_elo_url = f'{self.base_url}/api/get_elo/'
        self.update_elo_url = f'{self.base_url}/api/add_eval_result/'


class Config:
    def __init__(self, config_type="mini"):
        from cchess_alphazero.configs import _tables
        if config_type not in _tables.TYPES:
            raise RuntimeError('unknown config_type: %s' % (config_type))
        t = _tables.TYPES[config_type]
        self.type = config_type
        self.opts = Options()
        self.resource = ResourceConfig()
        self.internet = InternetConfig()
        self.model = ModelConfig(**t["model"])
        self.play = PlayConfig(**t["play"])
        self.play_data = PlayDataConfig(**t["play_data"])
        self.trainer = TrainerConfig(**t["trainer"])
        self.eval = EvaluateConfig(**t["eval"])
        self.engine = EngineConfig()
