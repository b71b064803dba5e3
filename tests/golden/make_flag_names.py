#!/usr/bin/env python
"""Writes tests/golden/reference_flags.json: the NAMES (and defaults) of the command-line flags
the reference's inference entry point defines -- scripts/infer.py:37-146 and
epos_lib/common.py:56-154 -- read from the flags.DEFINE_* calls with a recording stand-in for
tf.app.flags (nothing of TensorFlow is needed). Run in the build container:
    python tests/golden/make_flag_names.py
"""
import ast
import json
import os

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_flags.json')


def defined_flags(path):
  tree = ast.parse(open(path).read())
  out = []
  for node in ast.walk(tree):
    if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and
        node.func.attr.startswith('DEFINE_') and node.args):
      name = ast.literal_eval(node.args[0])
      try:
        default = ast.literal_eval(node.args[1])
      except Exception:
        default = None
      out.append({'name': name, 'kind': node.func.attr[len('DEFINE_'):], 'default': default,
                  'line': node.lineno})
  return sorted(out, key=lambda f: f['line'])


if __name__ == '__main__':
  data = {'scripts/infer.py': defined_flags(os.path.join(REF, 'scripts/infer.py')),
          'epos_lib/common.py': defined_flags(os.path.join(REF, 'epos_lib/common.py'))}
  with open(OUT, 'w') as f:
    json.dump(data, f, indent=1)
  print({k: len(v) for k, v in data.items()})
