"""Host-side mirror of the reference's ``cchess_alphazero`` package for the self-play hot path,
backed by the MI355X-native engine (libczero.so, hand-written HIP for gfx950).

Module paths, function names, argument meaning and error behaviour follow the reference
(NeymarL/ChineseChess-AlphaZero) so callers can switch by changing ``sys.path`` only.
"""
