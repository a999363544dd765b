import sys

from . import PRESETS, preset

if __name__ == '__main__':
    if len(sys.argv) >= 3 and sys.argv[1] == '--write':
        sys.stdout.write(preset(sys.argv[2]).dump())
    else:
        print('\n'.join(sorted(PRESETS)))
