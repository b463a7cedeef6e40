"""Per-type parameter values of the reference (configs/mini.py, configs/normal.py, configs/distribute.py),
kept as data.  `benchmark_*` are the synthetic settings of BASELINE.json."""

_model = dict(cnn_filter_num=256, cnn_first_filter_size=5, cnn_filter_size=3, res_layer_num=7, l2_reg=1e-4,
              value_fc_size=256, distributed=False, input_depth=14)

_play_mini = dict(max_processes=1, search_threads=10, vram_frac=1.0, simulation_num_per_move=100, c_puct=1.5,
                  noise_eps=0.25, dirichlet_alpha=0.2, tau_decay_rate=0.98, virtual_loss=3, max_game_length=100,
                  share_mtcs_info_in_self_play=False, reset_mtcs_info_per_game=5, enable_resign_rate=0.1,
                  resign_threshold=-0.92, min_resign_turn=20)
_play_normal = dict(_play_mini, max_processes=10, search_threads=40, simulation_num_per_move=800, thinking_loop=1,
                    logging_thinking=False, noise_eps=0.15, tau_decay_rate=0.9, resign_threshold=-0.98,
                    min_resign_turn=40, enable_resign_rate=0.5)
_play_distribute = dict(_play_normal, search_threads=10, c_puct=5, noise_eps=0.2, tau_decay_rate=0.9,
                        max_game_length=200, resign_threshold=-0.99, min_resign_turn=40, enable_resign_rate=0.99)

_eval_mini = dict(vram_frac=1.0, game_num=2, simulation_num_per_move=20, thinking_loop=1, c_puct=1,
                  tau_decay_rate=0, noise_eps=0.2, max_game_length=100, max_processes=2, search_threads=10)
_eval_normal = dict(_eval_mini, simulation_num_per_move=800, max_game_length=200, max_processes=10,
                    search_threads=8, next_generation_replace_rate=0.55)
_eval_distribute = dict({k: v for k, v in _eval_normal.items() if k != "next_generation_replace_rate"},
                        game_num=10, max_processes=10, search_threads=10, noise_eps=0.1, tau_decay_rate=0.5)

_trainer_mini = dict(min_games_to_begin_learn=1, min_data_size_to_learn=0, cleaning_processes=1, vram_frac=1.0,
                     batch_size=2, epoch_to_checkpoint=1, dataset_size=100000, start_total_steps=0,
                     save_model_steps=25, load_data_steps=100, momentum=0.9, loss_weights=[1.25, 1.0],
                     lr_schedules=[(0, 0.01), (150000, 0.001), (300000, 0.0001)], sl_game_step=10000, load_step=6)
_trainer_normal = dict({k: v for k, v in _trainer_mini.items() if k != "load_step"}, min_games_to_begin_learn=200, cleaning_processes=4, batch_size=512,
                       epoch_to_checkpoint=3, loss_weights=[1.0, 1.0],
                       lr_schedules=[(0, 0.01), (150000, 0.003), (400000, 0.0001)], sl_game_step=2000)
_trainer_distribute = dict(_trainer_normal, min_games_to_begin_learn=5000, cleaning_processes=20, batch_size=1024,
                           epoch_to_checkpoint=1, dataset_size=90000000, load_step=25000,
                           lr_schedules=[(0, 0.03), (100000, 0.01), (200000, 0.003), (300000, 0.001),
                                         (400000, 0.0003), (500000, 0.0001)])

TYPES = {
    "mini": dict(model=_model, play=_play_mini, eval=_eval_mini, trainer=_trainer_mini,
                 play_data=dict(sl_nb_game_in_file=250, nb_game_in_file=1, max_file_num=10, nb_game_save_record=1)),
    "normal": dict(model=_model, play=_play_normal, eval=_eval_normal, trainer=_trainer_normal,
                   play_data=dict(sl_nb_game_in_file=250, nb_game_in_file=5, max_file_num=300,
                                  nb_game_save_record=1)),
    "distribute": dict(model=dict(_model, cnn_filter_num=192, res_layer_num=10),
                       play=_play_distribute, eval=_eval_distribute, trainer=_trainer_distribute,
                       play_data=dict(sl_nb_game_in_file=250, nb_game_in_file=1, max_file_num=5000,
                                      nb_game_save_record=1)),
}


def benchmark_overrides(name):
    """BASELINE.json configs -> (model overrides, play overrides, engine overrides)."""
    normal_play = dict(simulation_num_per_move=800, c_puct=1.5, virtual_loss=3, dirichlet_alpha=0.2, noise_eps=0.15,
                       tau_decay_rate=0.9, max_game_length=100, search_threads=8)
    if name == "mini":
        return (dict(cnn_filter_num=32, res_layer_num=2),
                dict(simulation_num_per_move=50, noise_eps=0.25, tau_decay_rate=0.98, search_threads=10),
                dict(games_per_gpu=1))
    if name == "normal":
        return dict(cnn_filter_num=128, res_layer_num=7), normal_play, dict(games_per_gpu=4096)
    if name == "deep":
        return (dict(cnn_filter_num=256, res_layer_num=20), dict(normal_play, simulation_num_per_move=1600),
                dict(games_per_gpu=4096, net_dtype="float16"))
    if name == "eval":
        return (dict(cnn_filter_num=128, res_layer_num=7),
                dict(normal_play, simulation_num_per_move=400, noise_eps=0.2, tau_decay_rate=0, c_puct=1),
                dict(games_per_gpu=200))
    raise KeyError(name)
