"""UCI / UCCI front-end on the MI355X engine (reference: cchess_alphazero/uci.py:39-330).

Same command set, option names and output lines as the reference front-end:
    uci, isready, ucinewgame, setoption name {gpu|Threads} value N, position {fen F | startpos} [moves ...], fen F ...,
    go [depth N] [movetime MS | time MS] [infinite] [wtime MS] [btime MS], stop, quit
    -> "info depth D score S time T [pv ...] nps N" and "bestmove M [ponder P]"
`depth N` means N * 100 simulations, a clock (`wtime` / `btime`) caps the search at depth 30, `movetime` arms a timer
that stops the search 10 ms early (uci.py:211-236).  One search at a time; the search runs on its own thread and a
`stop` (or the timer) ends it through CChessPlayer.close_and_return_action.

    PYTHONPATH=chinesechess-alphazero_amd python -m cchess_alphazero.uci [--type mini|normal|distribute]
"""
import sys
import threading
from logging import getLogger
from time import time

from cchess_alphazero.environment import static_env as senv
from cchess_alphazero.environment.lookup_tables import flip_move

logger = getLogger(__name__)

ENGINE_ID = ("id name CCZero-MI355X", "id author https://cczero.org (reference), MI355X-native engine",
             "id version 2.4", "option name gpu spin default 0 min 0 max 7",
             "option name Threads spin default 10 min 0 max 1024")


class UCI:
    def __init__(self, config, out=None):
        self.config = config
        self.out = out or sys.stdout
        self.model = None
        self.pipe = None
        self.player = None
        self.is_ready = False
        self.use_history = False
        self.search_tree = {}
        self._lock = threading.Lock()          # one "bestmove" per "go"
        self._answered = True
        self._timer = None
        self._worker = None
        self._reset_position()

    # ---- plumbing ----
    def _say(self, line):
        print(line, file=self.out)
        self.out.flush()
        logger.debug(line)

    def _reset_position(self):
        self.state = senv.INIT_STATE
        self.history = [self.state]
        self.is_red_turn = True
        self.turns = 0

    def main(self, stream=None):
        for raw in (stream or sys.stdin):
            if self.handle(raw.strip()) is False:
                break

    def handle(self, line):
        """One command line; returns False on `quit`."""
        if not line:
            return True
        name, *args = line.split()
        fn = getattr(self, "cmd_" + name, None)
        if fn is None:
            logger.error(f"Error command: {line}")
            return True
        return fn(args)

    # ---- commands ----
    def cmd_uci(self, args):
        for line in ENGINE_ID:
            self._say(line)
        self._say("uciok")
        self.use_history = self.load_model()
        self.is_ready = True
        self._reset_position()

    def cmd_isready(self, args):
        if self.is_ready:
            self._say("readyok")

    def cmd_ucinewgame(self, args):
        self._reset_position()
        self.search_tree = {}
        self.is_ready = self.model is not None

    def cmd_setoption(self, args):
        # setoption name <id> value <x>
        if len(args) >= 4 and args[0] == "name" and args[2] == "value":
            if args[1] == "gpu":
                self.config.opts.device_list = args[3]
            elif args[1] == "Threads":
                self.config.play.search_threads = int(args[3])

    def cmd_position(self, args):
        if not self.is_ready:
            return
        moves_at = None
        if not args or args[0] == "startpos":
            self._reset_position()
            if len(args) > 1 and args[1] == "moves":
                moves_at = 2
        elif args[0] == "fen":
            # fen <position> <side> - - <halfmove> <fullmove> [moves ...]
            try:
                state = senv.fen_to_state(args[1])
                side, fullmove = args[2], int(args[6])
            except Exception as e:                        # malformed command: keep the previous position
                logger.error(f"cmd position error! cmd = {args}, {e}")
                return
            self.history = [state]
            if side == "b":
                self.state = senv.fliped_state(state)
                self.is_red_turn = False
                self.turns = (fullmove - 1) * 2 + 1
            else:
                self.state = state
                self.is_red_turn = True
                self.turns = (fullmove - 1) * 2
            if len(args) > 7 and args[7] == "moves":
                moves_at = 8
        elif args[0] == "moves":
            moves_at = 1
        if moves_at is not None:
            for tok in args[moves_at:]:
                mov = senv.parse_ucci_move(tok)
                if not self.is_red_turn:
                    mov = flip_move(mov)
                self.history.append(mov)
                self.state = senv.step(self.state, mov)
                self.is_red_turn = not self.is_red_turn
                self.turns += 1
                self.history.append(self.state)

    def cmd_fen(self, args):
        return self.cmd_position(["fen"] + args)

    def cmd_go(self, args):
        if not self.is_ready:
            return
        from cchess_alphazero.agent.player import CChessPlayer
        self._finish_worker()
        depth, infinite, budget = None, True, None
        for i, a in enumerate(args):
            if a == "depth":
                depth, infinite = int(args[i + 1]) * 100, False
            elif a in ("movetime", "time"):
                budget = int(args[i + 1]) / 1000
            elif a == "infinite":
                infinite = True
            elif (a == "wtime" and self.is_red_turn) or (a == "btime" and not self.is_red_turn):
                budget, depth, infinite = int(args[i + 1]) / 1000, 3000, False
        self.start_time = time()
        self.search_tree = {}
        self.player = CChessPlayer(self.config, search_tree=self.search_tree, pipes=self.pipe, enable_resign=False,
                                   debugging=True, uci=True, use_history=self.use_history, side=self.turns % 2)
        self.player.out = self.out
        self.player._idle.clear()                         # a `stop` that beats the worker to the player must wait for it
        with self._lock:
            self._answered = False
        self._worker = threading.Thread(target=self._search, args=(self.player, depth, infinite), daemon=True)
        self._worker.start()
        if budget:
            self._timer = threading.Timer(max(budget - 0.01, 0.0), self.cmd_stop, args=([],))
            self._timer.daemon = True
            self._timer.start()

    def cmd_stop(self, args):
        if not self.is_ready or self.player is None:
            return
        player = self.player
        ret = player.close_and_return_action(self.state, self.turns, self._repeated_replies(check_foul=False))
        self._worker_answer(player, ret)

    def cmd_ponderhit(self, args):
        pass                                              # the reference accepts and ignores it

    def cmd_quit(self, args):
        self._finish_worker()
        return False

    # ---- search ----
    def load_model(self):
        from cchess_alphazero.agent.model import CChessModel
        self.model = CChessModel(self.config)
        res = self.config.resource
        if not self.model.load(res.model_best_config_path, res.model_best_weight_path):
            self.model.build()
        self.pipe = self.model.get_pipes(need_reload=False)
        return self.model.model.cfg["input_depth"] == 28

    def _repeated_replies(self, check_foul):
        """Moves already played from the current position earlier in the game (uci.py:244-250, :281-288)."""
        if self.state not in self.history[:-1]:
            return None
        out = []
        for i in range(len(self.history) - 1):
            if self.history[i] == self.state:
                mov = self.history[i + 1]
                if not check_foul or senv.will_check_or_catch(self.state, mov):
                    out.append(mov)
        return out

    def _search(self, player, depth, infinite):
        no_act = None
        check = senv.done(self.state, need_check=True)[-1]
        if not check:
            no_act = self._repeated_replies(check_foul=True)
        action, _ = player.action(self.state, self.turns, no_act=no_act, depth=depth, infinite=infinite,
                                  hist=self.history)
        if action is None:
            self._worker_answer(player, None)
        else:
            self._worker_answer(player, (action, player.debug[self.state][1], player.done_tasks // 100))

    def _worker_answer(self, player, ret):
        with self._lock:
            if self._answered:
                return
            self._answered = True
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None
        if ret is None:
            self._say("bestmove none")
        else:
            self.info_best_move(*ret)
        if self.player is player:
            self.player = None

    def _finish_worker(self):
        if self._worker is not None and self._worker.is_alive() and self.player is not None:
            self.player.job_done = True
            self._worker.join(timeout=30)
        if self.player is not None:
            self.player.close(wait=False)
            self.player = None
        self._worker = None

    def info_best_move(self, action, value, depth):
        if not self.is_red_turn:
            value = -value
        duration = max(time() - self.start_time, 1e-9)
        self._say(f"info depth {depth} score {int(value * 1000)} time {int(duration * 1000)} "
                  f"nps {int(depth * 100 / duration) * 1000}")
        ponder = None
        node = self.search_tree.get(senv.step(self.state, action))
        if node is not None:
            best = 0
            for mov, a in node.a.items():
                if a.n > best:
                    ponder, best = mov, a.n
        if not self.is_red_turn:
            action = flip_move(action)
        line = f"bestmove {senv.to_uci_move(action)}"
        if ponder:
            line += f" ponder {senv.to_uci_move(flip_move(ponder) if self.is_red_turn else ponder)}"
        self._say(line)


def main(argv=None):
    import argparse
    from cchess_alphazero.config import Config, PlayWithHumanConfig
    ap = argparse.ArgumentParser()
    ap.add_argument("--type", default="distribute", choices=["mini", "normal", "distribute"])
    a = ap.parse_args(argv)
    config = Config(config_type=a.type)
    config.opts.device_list = "0"
    config.resource.create_directories()
    PlayWithHumanConfig().update_play_config(config.play)
    UCI(config).main()


if __name__ == "__main__":
    main()
