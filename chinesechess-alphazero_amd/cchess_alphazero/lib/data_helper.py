"""Play-record files (reference: cchess_alphazero/lib/data_helper.py and SelfPlayWorker.save_play_data,
worker/self_play.py:214-251).  The format is what the reference's ``opt`` trainer reads:
one JSON list ``[init_state, [move, value], [move, value], ...]`` per file (``nb_game_in_file`` games are
concatenated into one flat list, as the reference does)."""
import json
import os
from datetime import datetime, timedelta, timezone
from glob import glob
from logging import getLogger

logger = getLogger(__name__)


def get_game_data_filenames(rc):
    pattern = os.path.join(rc.play_data_dir, rc.play_data_filename_tmpl % "*")
    return list(sorted(glob(pattern)))


def write_game_data_to_file(path, data):
    with open(path, "wt") as f:
        json.dump(data, f)


def read_game_data_from_file(path):
    with open(path, "rt") as f:
        return json.load(f)


class PlayDataWriter:
    """Buffers finished games and writes ``play_<Beijing time>.json`` files (self_play.py:214-232);
    keeps at most ``play_data.max_file_num`` files (self_play.py:243-251)."""

    def __init__(self, config, rank=0, world=1):
        self.config = config
        self.rank, self.world = rank, world
        self.buffer = []
        self.idx = 1
        self._last_stamp = None
        self.files_written = 0
        os.makedirs(config.resource.play_data_dir, exist_ok=True)

    def _game_id(self):
        bj = datetime.utcnow().replace(tzinfo=timezone.utc).astimezone(timezone(timedelta(hours=8)))
        if self._last_stamp is not None and bj <= self._last_stamp:
            bj = self._last_stamp + timedelta(microseconds=1)
        self._last_stamp = bj
        s = bj.strftime("%Y%m%d-%H%M%S.%f")
        return s if self.world == 1 else f"{s}-r{self.rank}"

    def add_game(self, data):
        """data: [init_state, [move, value], ...] of one stored game."""
        self.buffer += data
        idx, self.idx = self.idx, self.idx + 1
        if idx % self.config.play_data.nb_game_in_file != 0:
            return None
        rc = self.config.resource
        path = os.path.join(rc.play_data_dir, rc.play_data_filename_tmpl % self._game_id())
        write_game_data_to_file(path, self.buffer)
        self.buffer = []
        self.files_written += 1
        self.remove_play_data()
        return path

    def remove_play_data(self):
        files = get_game_data_filenames(self.config.resource)
        extra = len(files) - self.config.play_data.max_file_num
        for f in files[:max(0, extra)]:
            try:
                os.remove(f)
            except OSError:
                pass
