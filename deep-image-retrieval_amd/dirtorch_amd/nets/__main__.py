from . import model_names
print('\n'.join(sorted(model_names)))
