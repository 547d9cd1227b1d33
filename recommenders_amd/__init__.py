"""MI355X-native candidate scoring + ranking evaluation (drop-in for the hot path of
Informfully/Recommenders, a Cornac fork).  See DESIGN.md and INTEGRATION.md."""
from .engine import DeviceCsr, HipScorer, IfrError  # noqa: F401
from .eval import UserResults, cache_rankings, clear_eval_cache, ranking_eval  # noqa: F401
from .exception import CornacException, ScoreException  # noqa: F401

__version__ = "0.1.0"
