from params.base_params import parse_base_args
from params.params_util import set_auto_params


def parse_train_params():
    return set_auto_params(parse_base_args("train"))
