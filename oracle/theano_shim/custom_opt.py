"""The reference's custom_opt.py only (re)registers a Theano graph optimisation; nothing to do in the shim."""
