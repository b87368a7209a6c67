"""Host-side mirror of the reference's ``representations`` package: same importable names, argument
meaning and error behaviour, every hot loop replaced by a libevrep call."""
