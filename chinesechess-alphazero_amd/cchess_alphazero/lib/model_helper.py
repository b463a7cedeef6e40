"""Best-model bookkeeping (reference: cchess_alphazero/lib/model_helper.py): load / save the best model and decide by
sha256 digest whether the weight file on disk is newer than the one in memory.  The download-from-server variant of the
reference (``load_best_model_weight_from_internet``) is part of its out-of-scope HTTP plumbing."""
from logging import getLogger

logger = getLogger(__name__)


def _best_paths(model):
    res = model.config.resource
    return res.model_best_config_path, res.model_best_weight_path


def load_best_model_weight(model):
    return model.load(*_best_paths(model))


def load_model_weight(model, config_path, weight_path, name=None):
    if name is not None:
        logger.info(f"loading {name} model weight")
    return model.load(config_path, weight_path)


def save_as_best_model(model):
    return model.save(*_best_paths(model))


def need_to_reload_best_model_weight(model):
    """True when the digest of the best-weight file differs from the digest of the weights in memory."""
    logger.debug("start reload the best model if changed")
    digest = model.fetch_digest(model.weight_file(_best_paths(model)[0], _best_paths(model)[1]))
    if digest is not None and digest != model.digest:
        return True
    logger.debug("the best model is not changed")
    return False
