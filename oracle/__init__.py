# TEST INFRASTRUCTURE ONLY: the CPU oracle.  Nothing under neuronika_amd/, host/ or include/ may import it.
