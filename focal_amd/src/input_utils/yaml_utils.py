import yaml


def load_yaml(file_path):
    with open(file_path, "r", errors="ignore") as stream:
        return yaml.safe_load(stream)
