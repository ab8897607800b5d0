"""`print_header` — banner used by eigsort / Focusr (`/root/reference/pyfocusr/main.py:1-6`)."""


def print_header(message, banner_length=72):
    bar = "=" * banner_length
    print(bar)
    print("")
    print(message)
    print("")
    print(bar)
